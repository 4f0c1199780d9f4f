"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding + all-gather logic of
gopairingbasedcryptography_amd/sharding.py with an oracle-backed stand-in for the GPU engine (tests may use the
oracle; the product module never does)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    from gopairingbasedcryptography_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 1 << 20, (1 << 18) + 3):
        for world in (1, 2, 4, 8):
            cuts = [shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


class OracleEngine:
    """Stand-in with the bn254 module's function names, computing on the CPU oracle."""
    def __init__(self):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_lib
        self.o = oracle_lib

    def pair_batch(self, P, Q): return self.o.pair_batch(P, Q)
    def g1_scalar_mul(self, b, k): return self.o.g1_scalar_mul(b, self._k(k))
    def g2_scalar_mul(self, b, k): return self.o.g2_scalar_mul(b, self._k(k))
    def g1_sum(self, p): return self.o.g1_sum(p)
    def g2_sum(self, p): return self.o.g2_sum(p)

    def multi_pair(self, P, Q, off): return self.o.multi_pair(P, Q, off)
    def gt_mul(self, a, b): return self.o.gt_mul(a, b)
    def gt_div(self, a, b): return self.o.gt_div(a, b)

    def _k(self, ks):
        import bn254_py as o
        return np.frombuffer(b"".join(o.scalar_to_bytes(int(k) % o.R) for k in ks), dtype=np.uint8) if isinstance(ks, (list, tuple)) else ks

    def gt_exp(self, x, k): return self.o.gt_exp(x, self._k(k))

    def pairing_check(self, P, Q):
        import bn254_py as o
        n = np.asarray(P).size // 64
        return self.o.multi_pair(P, Q, [0, n])[0].tobytes() == o.gt_to_bytes(o.F12_ONE)


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bn254_py as o
    from gopairingbasedcryptography_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = OracleEngine()
    g1 = np.frombuffer(o.g1_to_bytes(o.G1_GEN), dtype=np.uint8)
    g2 = np.frombuffer(o.g2_to_bytes(o.G2_GEN), dtype=np.uint8)
    sc = lambda tag, i0, m: np.frombuffer(b"".join(o.scalar_to_bytes(o.bench_scalar(tag, i0 + i)) for i in range(m)), dtype=np.uint8)
    lo, hi = sharding.shard_range(n, rank, world)
    # --- config 5 shape: sharded pair_batch + all-gather of GT
    P = eng.g1_scalar_mul(g1, sc("P", lo, hi - lo)); Q = eng.g2_scalar_mul(g2, sc("Q", lo, hi - lo))
    gathered = sharding.pair_batch_gather(eng, P, Q, n).numpy()
    Pall = eng.g1_scalar_mul(g1, sc("P", 0, n)); Qall = eng.g2_scalar_mul(g2, sc("Q", 0, n))
    ok_gather = bool((gathered == eng.pair_batch(Pall, Qall)).all())
    # --- config 3 shape: aggregate verify, valid and with one forged signature
    H = eng.g2_scalar_mul(g2, sc("H", 0, 1))[0]
    x = sc("x", lo, hi - lo)
    pk = eng.g1_scalar_mul(g1, x); sig = eng.g2_scalar_mul(H, x)
    rho = sc("rho", lo, hi - lo).copy().reshape(-1, 32); rho[:, 16:] = 0          # 128-bit verifier scalars
    neg = lambda b: np.frombuffer(o.g2_to_bytes(o.g2_neg(o.g2_from_bytes(b.tobytes()))), dtype=np.uint8)
    ok_valid = sharding.aggregate_verify(eng, pk, rho, sig, H, g1, neg)
    if rank == world - 1:
        sig = sig.copy(); sig[0] = eng.g2_scalar_mul(H, sc("forged", 0, 1))[0]
    ok_forged = sharding.aggregate_verify(eng, pk, rho, sig, H, g1, neg)
    # --- config 5 end to end: AFP25 batch decryption of the rank's shard of items, then the all-gather of the GT masks
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from afp25_fixture import Instance
    from gopairingbasedcryptography_amd import afp25
    inst = Instance(eng, B=4, n_items=3)                         # same instance on every rank (deterministic), 3 items: shards 2 + 1
    pis = [afp25.commit_g1(eng, inst.g1, inst.tau_powers, afp25.quotient_by_root(inst.f, it[0])) for it in inst.items]
    a, b = sharding.shard_range(len(inst.items), rank, world)
    sel = list(range(a, b))
    got = sharding.afp25_decrypt_gather(eng, np.stack([np.asarray(inst.D)] * len(sel)), np.stack([pis[i] for i in sel]),
                                        np.stack([np.asarray(inst.sk)] * len(sel)), np.stack([inst.items[i][1] for i in sel]),
                                        np.stack([inst.items[i][2] for i in sel]), len(inst.items)).numpy()
    ok_afp25 = bool((got == np.stack(inst.msgs)).all())
    q.put((rank, ok_gather, ok_valid, ok_forged, ok_afp25))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gather_and_aggregate_verify():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, n = 2, 7                      # ragged: shards of 4 and 3
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_gather, ok_valid, ok_forged, ok_afp25 in res:
        assert ok_gather, rank
        assert ok_valid is True, rank
        assert ok_forged is False, rank
        assert ok_afp25, rank
