"""Wall-clock time of the host-pointer entries over call sizes: looks for calls that cost far more than their size explains (a serial
chain on one lane, a kernel sized for large batches).  usage: python tools/size_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gopairingbasedcryptography_amd import bn254
g1, g2 = bn254.generators()
rng = np.random.default_rng(11)
N = 65536
k = rng.integers(0, 256, size=(N, 32), dtype=np.uint8); k[:, 31] &= 0x1f
P = bn254.g1_scalar_mul(g1, k.reshape(-1)); Q = bn254.g2_scalar_mul(g2, k[::-1].copy().reshape(-1))
GT = bn254.pair_batch(P[:4096], Q[:4096])
def t(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
sizes = (1, 16, 256, 4096, 65536)
rows = [
    ("g1_scalar_mul", lambda n: bn254.g1_scalar_mul(P[:n], k[:n].reshape(-1))),
    ("g2_scalar_mul", lambda n: bn254.g2_scalar_mul(Q[:n], k[:n].reshape(-1))),
    ("g1_sum", lambda n: bn254.g1_sum(P[:n])),
    ("g2_sum", lambda n: bn254.g2_sum(Q[:n])),
    ("g1_scalar_mul_sum", lambda n: bn254.g1_scalar_mul_sum(P[:n], k[:n].reshape(-1))),
    ("g2_scalar_mul_sum", lambda n: bn254.g2_scalar_mul_sum(Q[:n], k[:n].reshape(-1))),
    ("pair_batch", lambda n: bn254.pair_batch(P[:n], Q[:n])),
    ("multi_pair, one segment", lambda n: bn254.multi_pair(P[:n], Q[:n], np.array([0, n], dtype=np.uint64))),
    ("multi_pair, segments of 2", lambda n: bn254.multi_pair(P[:n], Q[:n], np.arange(0, n + 1, 2).astype(np.uint64)) if n > 1 else None),
    ("gt_exp", lambda n: bn254.gt_exp(GT[:n], k[:n].reshape(-1)) if n <= 4096 else None),
    ("gt_mul", lambda n: bn254.gt_mul(GT[:n], GT[:n]) if n <= 4096 else None),
    ("hash_to_g1 (32-byte messages)", lambda n: bn254.hash_to_g1([bytes(k[i]) for i in range(n)], b"sweep") if n <= 4096 else None),
    ("hash_to_g2 (32-byte messages)", lambda n: bn254.hash_to_g2([bytes(k[i]) for i in range(n)], b"sweep") if n <= 4096 else None),
    ("g1_unmarshal compressed", lambda n: bn254.g1_unmarshal(C1[:n * 32].copy(), elem_bytes=32)),
    ("g2_unmarshal compressed", lambda n: bn254.g2_unmarshal(C2[:n * 64].copy(), elem_bytes=64)),
]
C1 = np.asarray(bn254.g1_marshal(P, compressed=True)).reshape(-1); C2 = np.asarray(bn254.g2_marshal(Q, compressed=True)).reshape(-1)
print("%-34s" % "ms per call at n =" + "".join("%11d" % n for n in sizes))
for name, fn in rows:
    out = []
    for n in sizes:
        try:
            out.append("%11.2f" % t(lambda: fn(n)) if fn(n) is not None else "%11s" % "-")
        except Exception as e:
            out.append("%11s" % "error")
    print("%-34s" % name + "".join(out))
fb = bn254.FixedBase(P[:256])
for n in (1, 16, 256):
    kk = rng.integers(0, 256, size=(n * 256, 32), dtype=np.uint8); kk[:, 31] &= 0x1f
    print("fixed-base MSM over 256 bases, %4d sums: %.2f ms" % (n, t(lambda: fb.msm(kk.reshape(-1)))))
