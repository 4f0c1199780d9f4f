"""Exact count of the v_mad_i64_i32 the DEVICE code executes per unit, from the interval harness (tools/bounds_check.cpp compiles the
kernels' headers for the host; every product routine adds the MADs of its device form).  Writes profiles/executed_mads.json, which
bench.py reads for the `*_executed_mad_*` fields: the nominal roofline counts SURVEY's 136 MACs x 12 000 / 2 500 / 7 500 Fp-mul, this is
what actually issues.  usage: python tests/executed_mads.py   (CPU only; lives under tests/ because it takes its inputs from the oracle; tests/test_device_math_bounds.py checks the file stays true)"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path.insert(0, os.path.join(ROOT, "oracle"))
SO = os.path.join(ROOT, "tools", "libgpbc_bounds.so")


def harness():
    src = os.path.join(ROOT, "tools", "bounds_check.cpp")
    csrc = os.path.join(ROOT, "gopairingbasedcryptography_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(f) > os.path.getmtime(SO) for f in deps):
        subprocess.check_call(["g++", "-O2", "-pthread", "-std=c++17", "-DGPBC_BOUNDS", "-shared", "-fPIC", "-o", SO, src])
    hc = ctypes.CDLL(SO)
    hc.hc_mads_take.restype = ctypes.c_double
    return hc


def count(hc=None, n=4):
    import bn254_py as o
    hc = hc or harness()
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ks = [o.bench_scalar("mads", i) for i in range(3 * n)]
    P = np.frombuffer(b"".join(o.g1_to_bytes(o.g1_mul(o.G1_GEN, k)) for k in ks[:n]), dtype=np.uint8).copy()
    Q = np.frombuffer(b"".join(o.g2_to_bytes(o.g2_mul(o.G2_GEN, k)) for k in ks[n:2 * n]), dtype=np.uint8).copy()
    K = np.frombuffer(b"".join(int(k).to_bytes(32, "little") for k in ks[2 * n:]), dtype=np.uint8).copy()
    out = {}
    hc.hc_mads_take()
    f = np.zeros(n * 384, dtype=np.uint8)
    hc.hc_pair_lanes(vp(P), vp(Q), ctypes.c_size_t(n), vp(f), 0)                # line phase (one lane) + accumulator (lane pair)
    out["miller_loop"] = hc.hc_mads_take() / n
    g = np.zeros(n * 384, dtype=np.uint8)
    hc.hc_pair_lanes(vp(P), vp(Q), ctypes.c_size_t(n), vp(g), 1)
    out["pairing"] = hc.hc_mads_take() / n
    out["final_exp"] = out["pairing"] - out["miller_loop"]
    r1 = np.zeros(n * 64, dtype=np.uint8)
    hc.hc_g1_mul(vp(P), vp(K), ctypes.c_size_t(n), vp(r1))
    out["g1"] = hc.hc_mads_take() / n
    r2 = np.zeros(n * 128, dtype=np.uint8)
    hc.hc_g2_mul(vp(Q), vp(K), ctypes.c_size_t(n), vp(r2))
    out["g2"] = hc.hc_mads_take() / n
    return out


if __name__ == "__main__":
    c = count()
    doc = {"what": "v_mad_i64_i32 per unit executed by the device code (all lanes of a pairing's lane pair summed), counted by tools/bounds_check.cpp; "
                   "averages over 4 random inputs (the GLV / GLS loops and the Legendre-free paths are data-independent to within a few additions)",
           "nominal_mac_per_unit": {"pairing": 12000 * 136, "g1": 2500 * 136, "g2": 7500 * 136},
           "mads_per_unit": {k: round(v) for k, v in c.items()}}
    with open(os.path.join(ROOT, "profiles", "executed_mads.json"), "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc, indent=1))
