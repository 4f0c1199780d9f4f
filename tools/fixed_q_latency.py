"""Wall-clock time of a FEW ciphertext-sized products against one shared G2 list: multi_pair_fixed_q (lines of the list computed once)
against multi_pair on the replicated list.  usage: python tools/fixed_q_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gopairingbasedcryptography_amd import bn254
g1, g2 = bn254.generators()
rng = np.random.default_rng(3)
m = 513
kq = rng.integers(0, 256, size=(m, 32), dtype=np.uint8); kq[:, 31] &= 0x1f
Qs = bn254.g2_scalar_mul(g2, kq.reshape(-1))
for k in (1, 2, 4, 16, 64):
    kp = rng.integers(0, 256, size=(m * k, 32), dtype=np.uint8); kp[:, 31] &= 0x1f
    Ps = bn254.g1_scalar_mul(g1, kp.reshape(-1))
    off = np.arange(0, m * k + 1, m).astype(np.uint64)
    Qrep = np.tile(Qs, (k, 1))
    a = bn254.multi_pair_fixed_q(Ps, Qs); b = bn254.multi_pair(Ps, Qrep, off)
    assert (a == b).all()
    t0 = time.perf_counter()
    for _ in range(3): bn254.multi_pair_fixed_q(Ps, Qs)
    tf = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3): bn254.multi_pair(Ps, Qrep, off)
    tm = (time.perf_counter() - t0) / 3
    print("%3d segments x %d pairs: multi_pair_fixed_q %.2f ms   multi_pair %.2f ms" % (k, m, tf * 1e3, tm * 1e3))
