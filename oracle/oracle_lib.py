"""ctypes binding of oracle/libbn254_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (gopairingbasedcryptography_amd) never does.  Buffers are numpy uint8 arrays in
gnark-crypto in-memory layout (see bn254_oracle.h).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

G1_BYTES, G2_BYTES, GT_BYTES, SCALAR_BYTES, FP_BYTES = 64, 128, 384, 32, 32


def build(force=False):
    """Compile the C restatement with gcc (needs only oracle/*.c, *.h)."""
    so = os.path.join(_HERE, "libbn254_oracle.so")
    if force or not os.path.exists(so) or not os.path.exists(os.path.join(_HERE, "libbn254_oracle_count.so")):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    return so


def _load(counting=False):
    key = "count" if counting else "plain"
    if key not in _LIBS:
        build()
        name = "libbn254_oracle_count.so" if counting else "libbn254_oracle.so"
        lib = ctypes.CDLL(os.path.join(_HERE, name))
        lib.gpbc_oracle_fp_mul_count.restype = ctypes.c_uint64
        _LIBS[key] = lib
    return _LIBS[key]


def _buf(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _out(n, width):
    o = np.zeros((n, width), dtype=np.uint8)
    return o, o.ctypes.data_as(ctypes.c_void_p)


def pair_batch(P, Q, threads=1):
    P, pP = _buf(P); Q, pQ = _buf(Q)
    n = P.size // G1_BYTES
    assert Q.size // G2_BYTES == n
    out, po = _out(n, GT_BYTES)
    _load().gpbc_oracle_pair_batch(pP, pQ, ctypes.c_size_t(n), po, ctypes.c_int(threads))
    return out


def multi_pair(P, Q, seg_off, threads=1):
    P, pP = _buf(P); Q, pQ = _buf(Q)
    seg = np.ascontiguousarray(seg_off, dtype=np.uint64)
    k = seg.size - 1
    out, po = _out(k, GT_BYTES)
    _load().gpbc_oracle_multi_pair(pP, pQ, seg.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(k), po,
                                   ctypes.c_int(threads))
    return out


def miller_loop(P, Q, threads=1):
    P, pP = _buf(P); Q, pQ = _buf(Q)
    n = P.size // G1_BYTES
    out, po = _out(n, GT_BYTES)
    _load().gpbc_oracle_miller_loop(pP, pQ, ctypes.c_size_t(n), po, ctypes.c_int(threads))
    return out


def final_exp(F, threads=1):
    F, pF = _buf(F)
    n = F.size // GT_BYTES
    out, po = _out(n, GT_BYTES)
    _load().gpbc_oracle_final_exp(pF, ctypes.c_size_t(n), po, ctypes.c_int(threads))
    return out


def _scalar_mul(fn, width, base, scalars, threads):
    base, pb = _buf(base); scalars, ps = _buf(scalars)
    n = scalars.size // SCALAR_BYTES
    nbase = base.size // width
    assert nbase in (1, n)
    out, po = _out(n, width)
    fn(pb, ctypes.c_size_t(nbase), ps, ctypes.c_size_t(n), po, ctypes.c_int(threads))
    return out


def g1_scalar_mul(base, scalars, threads=1):
    return _scalar_mul(_load().gpbc_oracle_g1_scalar_mul, G1_BYTES, base, scalars, threads)


def g2_scalar_mul(base, scalars, threads=1):
    return _scalar_mul(_load().gpbc_oracle_g2_scalar_mul, G2_BYTES, base, scalars, threads)


def g1_sum(pts):
    pts, pp = _buf(pts)
    out, po = _out(1, G1_BYTES)
    _load().gpbc_oracle_g1_sum(pp, ctypes.c_size_t(pts.size // G1_BYTES), po)
    return out[0]


def g2_sum(pts):
    pts, pp = _buf(pts)
    out, po = _out(1, G2_BYTES)
    _load().gpbc_oracle_g2_sum(pp, ctypes.c_size_t(pts.size // G2_BYTES), po)
    return out[0]


def gt_exp(x, k, threads=1):
    x, px = _buf(x); k, pk = _buf(k)
    n = x.size // GT_BYTES
    out, po = _out(n, GT_BYTES)
    _load().gpbc_oracle_gt_exp(px, pk, ctypes.c_size_t(n), po, ctypes.c_int(threads))
    return out


def _gt_binary(fn, a, b):
    a, pa = _buf(a); b, pb = _buf(b)
    n = a.size // GT_BYTES
    out, po = _out(n, GT_BYTES)
    fn(pa, pb, ctypes.c_size_t(n), po)
    return out


def gt_mul(a, b): return _gt_binary(_load().gpbc_oracle_gt_mul, a, b)
def gt_div(a, b): return _gt_binary(_load().gpbc_oracle_gt_div, a, b)


def gt_inverse(a):
    a, pa = _buf(a)
    n = a.size // GT_BYTES
    out, po = _out(n, GT_BYTES)
    _load().gpbc_oracle_gt_inverse(pa, ctypes.c_size_t(n), po)
    return out


def fp_mul(a, b):
    a, pa = _buf(a); b, pb = _buf(b)
    n = a.size // FP_BYTES
    out, po = _out(n, FP_BYTES)
    _load().gpbc_oracle_fp_mul(pa, pb, ctypes.c_size_t(n), po)
    return out


def fp_inv(a):
    a, pa = _buf(a)
    n = a.size // FP_BYTES
    out, po = _out(n, FP_BYTES)
    _load().gpbc_oracle_fp_inv(pa, ctypes.c_size_t(n), po)
    return out


def fp12_cyclotomic_square(a):
    a, pa = _buf(a)
    n = a.size // GT_BYTES
    out, po = _out(n, GT_BYTES)
    _load().gpbc_oracle_fp12_cyclotomic_square(pa, ctypes.c_size_t(n), po)
    return out


def fp_mul_counts():
    """Fp-mul counts per op from the instrumented build (single thread): dict of op -> count."""
    lib = _load(counting=True)
    import sys
    sys.path.insert(0, _HERE)
    import bn254_py as o
    P = np.frombuffer(o.g1_to_bytes(o.g1_mul(o.G1_GEN, 12345)), dtype=np.uint8)
    Q = np.frombuffer(o.g2_to_bytes(o.g2_mul(o.G2_GEN, 67890)), dtype=np.uint8)
    k = np.frombuffer(o.scalar_to_bytes(o.bench_scalar("count", 0)), dtype=np.uint8)
    res = {}
    f = np.zeros(GT_BYTES, dtype=np.uint8); gt = np.zeros(GT_BYTES, dtype=np.uint8)
    g1 = np.zeros(G1_BYTES, dtype=np.uint8); g2 = np.zeros(G2_BYTES, dtype=np.uint8)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    one = ctypes.c_size_t(1)
    lib.gpbc_oracle_fp_mul_count(1)
    lib.gpbc_oracle_miller_loop(vp(P), vp(Q), one, vp(f), ctypes.c_int(1))
    res["miller_loop"] = lib.gpbc_oracle_fp_mul_count(1)
    lib.gpbc_oracle_final_exp(vp(f), one, vp(gt), ctypes.c_int(1))
    res["final_exp"] = lib.gpbc_oracle_fp_mul_count(1)
    lib.gpbc_oracle_g1_scalar_mul(vp(P), one, vp(k), one, vp(g1), ctypes.c_int(1))
    res["g1_scalar_mul"] = lib.gpbc_oracle_fp_mul_count(1)
    lib.gpbc_oracle_g2_scalar_mul(vp(Q), one, vp(k), one, vp(g2), ctypes.c_int(1))
    res["g2_scalar_mul"] = lib.gpbc_oracle_fp_mul_count(1)
    lib.gpbc_oracle_gt_exp(vp(gt), vp(k), one, vp(f), ctypes.c_int(1))
    res["gt_exp"] = lib.gpbc_oracle_fp_mul_count(1)
    return res
