#!/usr/bin/env python3
"""Per-kernel times (gpbc_profile_begin / _end) of the BSW07 and AFP25 decrypt legs of bench.py at 1/4 of the BASELINE sizes.
    python tools/leg_profile.py"""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gopairingbasedcryptography_amd import _lib, afp25, bn254, bsw07
import bench_workloads as wl
bn254.init(0)
lib = _lib.load()
dev = torch.device("cuda", 0)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

def profiled(name, fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    _lib.check(lib.gpbc_profile_begin(stream))
    fn()
    names = ctypes.create_string_buffer(32 * 32); ms = (ctypes.c_double * 32)(); cnt = (ctypes.c_int * 32)(); nk = ctypes.c_int(0)
    _lib.check(lib.gpbc_profile_end(names, ms, cnt, 32, ctypes.byref(nk)))
    print("%s: %.2f ms between the stream's profile marks, %.2f ms wall clock for the call" % (name, sum(ms[i] for i in range(nk.value)), wall))
    for i in range(nk.value):
        print("   %-30s %9.3f ms  x%d" % (names.raw[32 * i:32 * i + 32].split(b"\0")[0].decode(), ms[i], cnt[i]))

n = 1 << 14
inst = wl.bsw07_instance(bn254, "256of256", n, dev)
folded = bsw07.fold_key(bn254, bsw07.decrypt_plan(inst["tree"], inst["attrs"]), inst["dj"], inst["dj_prime"])
profiled("BSW07 decrypt, %d ciphertexts x 513 pairs" % n, lambda: bsw07.decrypt_batch_arrays(bn254, folded, inst["D"], inst["c_tilde"], inst["c"], inst["cy"], inst["cy_prime"]))
del inst
m = 1 << 16
a = wl.afp25_instance(bn254, 256, m, dev)
profiled("AFP25 decrypt, %d items x 3 pairs" % m, lambda: afp25.decrypt_batch_arrays(bn254, a["D"], a["pi"], a["sk"], a["C1"], a["C2"]))
