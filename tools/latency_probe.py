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
# the same with the points already in HBM (device tensors in, device tensor out, one synchronisation): what the kernels alone cost
import torch
for n in (1, 2, 64):
    dP = torch.from_numpy(np.repeat(g1[None], n, 0)).cuda(); dQ = torch.from_numpy(np.repeat(g2[None], n, 0)).cuda()
    out = torch.empty((n, 384), dtype=torch.uint8, device="cuda")
    bn254.pair_batch(dP, dQ, out=out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        bn254.pair_batch(dP, dQ, out=out); torch.cuda.synchronize()
    print("pair_batch device-resident", n, "%.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
    f = bn254.miller_loop(dP, dQ); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f = bn254.miller_loop(dP, dQ); torch.cuda.synchronize()
    print("  miller_loop alone", n, "%.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
    t0 = time.perf_counter()
    for _ in range(10):
        e = bn254.final_exp(f); torch.cuda.synchronize()
    print("  final_exp alone", n, "%.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
