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

# the other single calls the reference makes one at a time: what ONE call costs through the host-pointer API
rng = np.random.default_rng(7)
k1 = bn254.scalars_to_bytes([int.from_bytes(rng.bytes(31), "little")])
gt1 = bn254.pair_batch(g1[None], g2[None])
def timed(name, fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    print("%-44s %.3f ms" % (name, (time.perf_counter() - t0) / reps * 1e3))
timed("g1_scalar_mul, 1 point", lambda: bn254.g1_scalar_mul(g1[None], k1))
timed("g2_scalar_mul, 1 point", lambda: bn254.g2_scalar_mul(g2[None], k1))
timed("g1_scalar_mul_base, 1 scalar", lambda: bn254.g1_scalar_mul_base(k1))
timed("g2_scalar_mul_base, 1 scalar", lambda: bn254.g2_scalar_mul_base(k1))
timed("gt_exp, 1 element", lambda: bn254.gt_exp(gt1, k1))
timed("gt_mul, 1 element", lambda: bn254.gt_mul(gt1, gt1))
timed("gt_div, 1 element", lambda: bn254.gt_div(gt1, gt1))
timed("hash_to_g1, 1 message of 32 bytes", lambda: bn254.hash_to_g1([b"m" * 32], b"Hash Bytes To Element In G1"))
timed("hash_to_g2, 1 message of 32 bytes", lambda: bn254.hash_to_g2([b"m" * 32], b"Hash Bytes To Element In G2"))
for m in (3, 513):
    P = np.repeat(g1[None], m, 0); Q = np.repeat(g2[None], m, 0)
    timed("multi_pair, 1 segment of %d pairs" % m, lambda: bn254.multi_pair(P, Q, np.array([0, m], dtype=np.uint64)))
