"""Wall-clock latency of small host-pointer calls (Pair / PairingCheck as the reference makes them: one at a time).  usage: python tools/latency_probe.py"""
import time, numpy as np
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gopairingbasedcryptography_amd import bn254
g1, g2 = bn254.generators()
for n in (1, 2, 64, 1024, 2048, 4096):
    P = np.repeat(g1[None], n, 0); Q = np.repeat(g2[None], n, 0)
    bn254.pair_batch(P, Q)
    t0 = time.perf_counter()
    for _ in range(5): bn254.pair_batch(P, Q)
    print("pair_batch", n, "%.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
P = np.repeat(g1[None], 2, 0); Q = np.repeat(g2[None], 2, 0)
bn254.pairing_check(P, Q)
t0 = time.perf_counter()
for _ in range(5): bn254.pairing_check(P, Q)
print("pairing_check 2 pairs %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
