#!/usr/bin/env python3
"""Per-kernel times of the bucket multi-scalar multiplication (gpbc_profile_begin / _end around gpbc_g1/g2_scalar_mul_sum_dev).
    python tools/msm_profile.py [log2_n] [scalar_bits]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gopairingbasedcryptography_amd import _lib, bn254
import bench_workloads as wl
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 256
bn254.init(0)
lib = _lib.load()
dev = torch.device("cuda", 0)
g1, g2 = bn254.generators()
d = lambda a: torch.from_numpy(np.array(a, dtype=np.uint8, copy=True)).to(dev)
kb = d(wl.bench_scalars("P", 0, n)).reshape(n, 32)
ks = d(wl.bench_scalars("s", 0, n)).reshape(n, 32).clone()
if bits < 256:
    ks[:, bits // 8:] = 0
P, Q = bn254.g1_scalar_mul(d(g1), kb), bn254.g2_scalar_mul(d(g2), kb)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, fn, X in (("G1", bn254.g1_scalar_mul_sum, P), ("G2", bn254.g2_scalar_mul_sum, Q)):
    fn(X, ks); torch.cuda.synchronize()
    _lib.check(lib.gpbc_profile_begin(stream))
    fn(X, ks)
    names = ctypes.create_string_buffer(32 * 32); ms = (ctypes.c_double * 32)(); cnt = (ctypes.c_int * 32)(); nk = ctypes.c_int(0)
    _lib.check(lib.gpbc_profile_end(names, ms, cnt, 32, ctypes.byref(nk)))
    tot = sum(ms[i] for i in range(nk.value))
    print("%s MSM, %d terms, %d-bit scalars: %.3f ms in kernels = %.1f M terms/s" % (name, n, bits, tot, n / tot / 1e3))
    for i in range(nk.value):
        print("   %-24s %8.3f ms  x%d" % (names.raw[32 * i:32 * i + 32].split(b"\0")[0].decode(), ms[i], cnt[i]))
