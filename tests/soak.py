#!/usr/bin/env python3
"""Randomised soak of the GPU engine against the C restatement (oracle/): many more inputs than the test-suite uses, to
reach rare paths (exact zero tests, exceptional additions, large table indices).  Test infrastructure (it imports the oracle, so it
lives under tests/; not collected by pytest) — run on a GPU box:
    python tests/soak.py [n [seed]]        (default n = 65536, seed 20261004)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_lib  # noqa: E402
from gopairingbasedcryptography_amd import bn254  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
threads = min(len(os.sched_getaffinity(0)), 16)
oracle_lib.build()
bn254.init(0)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261004
rng = np.random.default_rng(seed)


def scal(m, full=False):
    a = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
    if not full:
        a[:, 31] &= 0x1f
    a[: min(m, 8)] = 0
    a[0, 0], a[1, 0], a[2, 0] = 1, 2, 3                       # tiny scalars among the first rows; row 3.. stay zero
    return a.reshape(-1)


g1, g2 = bn254.generators()
t0 = time.time()
P = bn254.g1_scalar_mul(g1, scal(n))                          # fixed-base path (n >= 16384)
Q = bn254.g2_scalar_mul(g2, scal(n))
k1, k2 = scal(n, full=True), scal(n, full=True)
R1 = bn254.g1_scalar_mul(P, k1)                               # variable base, 256-bit scalars
R2 = bn254.g2_scalar_mul(Q, k2)
assert (R1 == oracle_lib.g1_scalar_mul(P, k1, threads=threads)).all(), "G1 scalar mul"
assert (R2 == oracle_lib.g2_scalar_mul(Q, k2, threads=threads)).all(), "G2 scalar mul"
print("scalar multiplications ok  (%d each, %.1f s)" % (n, time.time() - t0), flush=True)
t0 = time.time()
gt = bn254.pair_batch(P, Q)
assert (gt == oracle_lib.pair_batch(P, Q, threads=threads)).all(), "pairing"
if n > 16384:                                                 # the same pairings again in small calls: the pipelined one-launch Miller loop
    want = gt.reshape(-1, 384)
    for lo in range(0, n, 8192 + 37):
        hi = min(n, lo + 8192 + 37)
        assert (bn254.pair_batch(P.reshape(-1, 64)[lo:hi], Q.reshape(-1, 128)[lo:hi]).reshape(-1, 384) == want[lo:hi]).all(), "pipelined small batch"
# ... and in calls of 1 .. 2048 pairs: the latency form (one pairing per wavefront, csrc/wide29.hip.hpp), sizes drawn at random
want = gt.reshape(-1, 384)
lo = 0
n_wide_calls = 0
while lo < min(n, 40000):
    hi = min(n, lo + int(rng.integers(1, 2049)))
    assert (bn254.pair_batch(P.reshape(-1, 64)[lo:hi], Q.reshape(-1, 128)[lo:hi]).reshape(-1, 384) == want[lo:hi]).all(), ("latency form", lo, hi)
    lo = hi
    n_wide_calls += 1
print("pairings ok  (%d, %.1f s; %d calls through the latency form)" % (n, time.time() - t0, n_wide_calls), flush=True)
t0 = time.time()
lens = rng.integers(0, 12, size=n // 4)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
m = int(off[-1])
idx = rng.integers(0, n, size=m)
Pm, Qm = P[idx], Q[idx]
assert (bn254.multi_pair(Pm, Qm, off) == oracle_lib.multi_pair(Pm, Qm, off, threads=threads)).all(), "multi-pairing"
# small multi-pairing calls (segment product + final exponentiation per wavefront): a few hundred pairs per call
for trial in range(24):
    ks = int(rng.integers(1, 40))
    ls = rng.integers(0, 30, size=ks)
    o2 = np.concatenate([[0], np.cumsum(ls)]).astype(np.uint64)
    ii = rng.integers(0, n, size=int(o2[-1]))
    if int(o2[-1]) == 0:
        continue
    assert (bn254.multi_pair(P[ii], Q[ii], o2) == oracle_lib.multi_pair(P[ii], Q[ii], o2, threads=threads)).all(), ("small multi-pairing", trial)
# ... and calls whose segments are long enough to be folded over 8 / 16 wavefronts first (k_segment_fold_wide): up to 2 048 pairs per call
for trial in range(16):
    ks = int(rng.integers(1, 9))
    cuts = np.sort(rng.integers(0, 2049, size=ks - 1)) if ks > 1 else np.zeros(0, dtype=np.int64)
    o2 = np.concatenate([[0], cuts, [int(rng.integers(max(int(cuts[-1]) if ks > 1 else 1, 1), 2049))]]).astype(np.uint64)
    ii = rng.integers(0, n, size=int(o2[-1]))
    assert (bn254.multi_pair(P[ii], Q[ii], o2) == oracle_lib.multi_pair(P[ii], Q[ii], o2, threads=threads)).all(), ("folded multi-pairing", trial, o2)
# ... a few LONG segments through the throughput kernels (k_segment_fold before the one-thread products): 5 000-pair calls over Miller
# values, and — with more than 131 072 pairs, where segments are cut into shared-squaring chunks — over chunk values
for sizes in ([1500, 3000, 1, 700], [5000], [70000, 0, 90000] if n >= 160000 else [30000, 0, 20000]):
    o2 = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    ii = rng.integers(0, n, size=int(o2[-1]))
    assert (bn254.multi_pair(P[ii], Q[ii], o2) == oracle_lib.multi_pair(P[ii], Q[ii], o2, threads=threads)).all(), ("long segments", sizes)
print("multi-pairings ok  (%d segments, %d pairs, %.1f s)" % (len(lens), m, time.time() - t0), flush=True)
t0 = time.time()
ke = scal(4096, full=True)
assert (bn254.gt_exp(gt[:4096], ke) == oracle_lib.gt_exp(gt[:4096], ke, threads=threads)).all(), "GT exp"
for lo_ in range(0, 4096, 1500):                              # the same in calls of at most 2 048 elements: one element per wavefront (k_gt_exp_wide)
    hi_ = min(4096, lo_ + 1500)
    kk = ke.reshape(-1, 32)[lo_:hi_].reshape(-1)
    assert (bn254.gt_exp(gt[lo_:hi_], kk) == oracle_lib.gt_exp(gt[lo_:hi_], kk, threads=threads)).all(), "GT exp, latency form"
assert (bn254.gt_mul(gt[:4096], gt[4096:8192]) == oracle_lib.gt_mul(gt[:4096], gt[4096:8192])).all(), "GT mul"
fb = bn254.FixedBase(P[:16])
ks = scal(16 * 512, full=True)
want = np.stack([np.asarray(oracle_lib.g1_sum(oracle_lib.g1_scalar_mul(P[:16], ks.reshape(512, -1)[i], threads=threads))).reshape(-1) for i in range(512)])
assert (fb.msm(ks) == want).all(), "fixed-base MSM"
back, ok = bn254.g2_unmarshal(bn254.g2_marshal(R2, compressed=True), elem_bytes=64)
assert ok.all() and (back == R2).all(), "G2 wire round trip"
back, ok = bn254.g1_unmarshal(bn254.g1_marshal(R1, compressed=True), elem_bytes=32)
assert ok.all() and (back == R1).all(), "G1 wire round trip"
print("GT ops, fixed-base MSM, wire round trips ok  (%.1f s)" % (time.time() - t0), flush=True)
t0 = time.time()
from gopairingbasedcryptography_amd import _lib, hash_to  # noqa: E402
lib = _lib.load()
try:
    for trial in range(12):                                   # fixed-Q multi-pairing: random list lengths, segment counts and chunk lengths
        mq, kq = int(rng.integers(1, 90)), int(rng.integers(1, 40))
        Qs = Q[rng.integers(0, n, size=mq)].copy()
        Ps = P[rng.integers(0, n, size=mq * kq)].copy()
        Qs[rng.integers(0, mq)] = 0 if trial % 3 == 0 else Qs[rng.integers(0, mq)]
        Ps[rng.integers(0, mq * kq)] = 0
        _lib.check(lib.gpbc_set_multi_pair_chunk(int(rng.integers(0, 65))))
        want = oracle_lib.multi_pair(Ps, np.tile(Qs, (kq, 1)), np.arange(0, mq * kq + 1, mq).astype(np.uint64), threads=threads)
        assert (bn254.multi_pair_fixed_q(Ps, Qs) == want).all(), ("fixed-Q multi-pairing", mq, kq)
finally:
    lib.gpbc_set_multi_pair_chunk(0)
print("fixed-Q multi-pairings ok  (%.1f s)" % (time.time() - t0), flush=True)
t0 = time.time()
msgs = [rng.bytes(int(rng.integers(0, 300))) for _ in range(4096)]
mont = lambda v: (v * (1 << 256) % hash_to.P_MOD).to_bytes(32, "little")
for dst in (b"soak", rng.bytes(255)):
    want = np.frombuffer(b"".join(mont(v) for m_ in msgs for v in hash_to.hash_to_field(m_, dst, 4)), dtype=np.uint8).reshape(len(msgs), 128)
    assert (bn254.hash_to_field(msgs, dst, 4) == want).all(), "hash_to_field"
H1, H2 = bn254.hash_to_g1(msgs, b"soak"), bn254.hash_to_g2(msgs, b"soak")
assert (H1 == hash_to.hash_to_g1_via_host_fields(msgs, b"soak")).all() and (H2 == hash_to.hash_to_g2_via_host_fields(msgs, b"soak")).all(), "hash to curve"
back, ok = bn254.g2_unmarshal(bn254.g2_marshal(H2))
assert ok.all() and (back == H2).all(), "hashed G2 points are in the group"
print("hash to field / curve ok  (%d messages, %.1f s)" % (len(msgs), time.time() - t0), flush=True)
print("soak OK  (n = %d, seed = %d)" % (n, seed))
