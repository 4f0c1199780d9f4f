#!/usr/bin/env python3
"""Rates of the host-pointer entries (pageable numpy buffers in, pageable out) against the HBM-resident ones, several repeats on
one box: pairings, G1 and G2 scalar multiplications at 2^20.   python tools/host_path_bench.py [repeats]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gopairingbasedcryptography_amd import bn254
import bench_workloads as wl
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bn254.init(0)
dev = torch.device("cuda", 0)
n = 1 << 20
g1, g2 = bn254.generators()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
kP, kQ, ks = (d(wl.bench_scalars(t, 0, n)).reshape(n, 32) for t in ("P", "Q", "s"))
P, Q = bn254.g1_scalar_mul(d(g1), kP), bn254.g2_scalar_mul(d(g2), kQ)
Ph, Qh, kh = P.cpu().numpy(), Q.cpu().numpy(), ks.cpu().numpy()
outs = {"pair": np.zeros((n, 384), np.uint8), "g1": np.zeros((n, 64), np.uint8), "g2": np.zeros((n, 128), np.uint8)}
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return time.perf_counter() - t0
cases = [("pairings", lambda: bn254.pair_batch(P, Q), lambda: bn254.pair_batch(Ph, Qh, out=outs["pair"])),
         ("g1 scalar mults", lambda: bn254.g1_scalar_mul(P, ks), lambda: bn254.g1_scalar_mul(Ph, kh)),
         ("g2 scalar mults", lambda: bn254.g2_scalar_mul(Q, ks), lambda: bn254.g2_scalar_mul(Qh, kh))]
for name, devfn, hostfn in cases:
    devfn(); hostfn()
    dv = min(t(devfn) for _ in range(reps)); hs = sorted(t(hostfn) for _ in range(reps))
    print("%-16s HBM-resident %7.2f M/s   host pointers best %7.2f M/s  median %7.2f M/s  (%.0f %%)" % (name, n / dv / 1e6, n / hs[0] / 1e6, n / hs[len(hs) // 2] / 1e6, 100 * dv / hs[0]))
