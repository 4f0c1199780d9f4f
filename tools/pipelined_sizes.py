#!/usr/bin/env python3
"""Pairings/s of HBM-resident batches of several sizes with the pipelined small-batch Miller loop on and off.
    python tools/pipelined_sizes.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gopairingbasedcryptography_amd import _lib, bn254
import bench_workloads as wl
bn254.init(0)
lib = _lib.load()
dev = torch.device("cuda", 0)
n = 1 << 17
g1, g2 = bn254.generators()
d = lambda a: torch.from_numpy(np.array(a, dtype=np.uint8, copy=True)).to(dev)
P = bn254.g1_scalar_mul(d(g1), d(wl.bench_scalars("P", 0, n)).reshape(n, 32))
Q = bn254.g2_scalar_mul(d(g2), d(wl.bench_scalars("Q", 0, n)).reshape(n, 32))
def rate(m, reps):
    p, q = P[:m].contiguous(), Q[:m].contiguous()
    bn254.pair_batch(p, q); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): bn254.pair_batch(p, q)
    torch.cuda.synchronize()
    return m * reps / (time.perf_counter() - t0)
for m in (64, 1024, 4096, 8192, 16384, 32768):
    out = []
    for mode in (0, 1):
        _lib.check(lib.gpbc_set_pipelined_miller(mode))
        out.append(rate(m, 8))
    print("%6d pairs: two kernels %8.3f M/s (%.2f ms per call)   pipelined %8.3f M/s (%.2f ms per call)" % (m, out[0] / 1e6, 1e3 * m / out[0], out[1] / 1e6, 1e3 * m / out[1]))
lib.gpbc_set_pipelined_miller(1)
