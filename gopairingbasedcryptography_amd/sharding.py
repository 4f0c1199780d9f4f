"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  Every pairing / scalar multiplication is an independent unit, so batches shard by contiguous index
range with NO data-path collective; a collective appears only where a combined result is needed:

  * aggregate_verify  (BASELINE config 3): each rank reduces its shard to partial sums A_g in G1 (64 B) and B_g in G2
    (128 B), one all-gather of 192 B per rank, then every rank adds the partials and runs the 2-pairing check;
  * pair_batch_gather / afp25_decrypt_gather (BASELINE config 5): all-gather of the per-shard GT values (n/G x 384 B per
    rank) — plain pairings, or the AFP25 batch decryption (3-pair multi-pairing + GT.Div per item) whose masks the config names.

`engine` is the compute backend: the `bn254` module of this package on a GPU box.  The CPU tests pass an
oracle-backed stand-in with the same function names — this module itself never imports the oracle.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n units owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _world():
    return dist.get_world_size() if dist.is_initialized() else 1


def _rank():
    return dist.get_rank() if dist.is_initialized() else 0


def init_library_comm(engine):
    """Give the engine's C library its own RCCL communicator spanning the torch.distributed job (one process per GPU): rank 0
    draws the id, torch.distributed carries the 128 bytes, every rank joins with its bound device.  Afterwards
    engine.allgather / engine.g1_scalar_mul_sum run their collective inside the library (ncclAllGather on bytes over xGMI)."""
    world, rank = _world(), _rank()
    ident = [engine.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(ident, src=0)
    engine.comm_init_rank(ident[0], world, rank)


def all_gather_rows(local, n_total, device=None, engine=None):
    """All-gather row blocks of a uint8 [n_local, width] array whose sizes follow shard_range(n_total, ...).

    One collective on a padded byte buffer: ncclAllGather on uint8 over xGMI — through the engine's own communicator when it
    has one (`engine.comm_ranks() == world`, see init_library_comm), through torch.distributed otherwise (gloo in the CPU
    tests)."""
    world = _world()
    t = torch.as_tensor(np.ascontiguousarray(local)) if not isinstance(local, torch.Tensor) else local
    if world == 1:
        return t
    if device is not None:
        t = t.to(device)
    width = t.shape[1]
    max_rows = (n_total + world - 1) // world
    pad = torch.zeros((max_rows, width), dtype=torch.uint8, device=t.device)
    pad[: t.shape[0]] = t
    if engine is not None and t.is_cuda and getattr(engine, "comm_ranks", lambda: 0)() == world:
        out = engine.allgather(pad.reshape(-1)).reshape(world * max_rows, width)
    else:
        out = torch.empty((world * max_rows, width), dtype=torch.uint8, device=t.device)
        dist.all_gather_into_tensor(out, pad)
    rows = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        rows.append(out[r * max_rows: r * max_rows + (hi - lo)])
    return torch.cat(rows, dim=0)


def pair_batch_gather(engine, P_local, Q_local, n_total, device=None):
    """Config 5 shape: every rank pairs its own shard, then all ranks receive all n_total GT values."""
    gt = engine.pair_batch(P_local, Q_local)
    return all_gather_rows(gt, n_total, device=device, engine=engine)


def afp25_decrypt_gather(engine, D_local, pi_local, sk_local, C1_local, C2_local, n_total, device=None):
    """BASELINE config 5 end to end: every rank decrypts its shard of the (ciphertext, identity) items —
    m_i = C2_i / (e(D_i, C1_i[0]) e(pi_i, C1_i[1]) e(sk_i, C1_i[2])), bibe/afp25_bibe/afp25_bibe.go:395-413, through
    afp25.decrypt_batch_arrays: one 3-pair multi-pairing and one GT.Div per item — and all ranks receive all n_total GT
    values by ONE all-gather (the library's RCCL communicator on the GPU box, gloo in the CPU tests)."""
    from . import afp25
    out = afp25.decrypt_batch_arrays(engine, D_local, pi_local, sk_local, C1_local, C2_local)
    return all_gather_rows(out, n_total, device=device, engine=engine)


def aggregate_verify(engine, pk_local, rho_local, sigma_local, H, g1, neg, device=None):
    """BLS aggregate verification with random linear combination.

    pk_local [m,64], sigma_local [m,128]: this rank's shard of public keys / signatures on the common message
    point H; rho_local [m,32]: the verifier's random scalars.  Checks  e(sum rho_i pk_i, H) * e(g1, -sum rho_i sigma_i) == 1.
    `neg` negates an affine G2 point on the host (field negation, gnark's G2Affine.Neg).  Returns bool (same on every rank).
    """
    if isinstance(pk_local, torch.Tensor) and pk_local.is_cuda and getattr(engine, "comm_ranks", lambda: 0)() == _world() and _world() > 1:
        # HBM-resident shards and a library communicator over the job: local sums, the all-gather of one point per rank and the
        # sum of the partials all happen inside the C library (gpbc_g1/g2_scalar_mul_sum_dev)
        A_all = engine.g1_scalar_mul_sum(pk_local, rho_local).cpu().numpy()
        B_all = engine.g2_scalar_mul_sum(sigma_local, rho_local).cpu().numpy()
        P = np.concatenate([A_all.reshape(-1), np.asarray(g1).reshape(-1)])
        Q = np.concatenate([np.asarray(H).reshape(-1), neg(B_all.reshape(-1))])
        return bool(engine.pairing_check(P, Q))
    A = engine.g1_sum(engine.g1_scalar_mul(pk_local, rho_local))
    B = engine.g2_sum(engine.g2_scalar_mul(sigma_local, rho_local))
    A = A if isinstance(A, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(A))
    B = B if isinstance(B, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(B))
    part = torch.cat([A.reshape(-1), B.reshape(-1)]).reshape(1, 192)
    world = _world()
    parts = all_gather_rows(part, world, device=device).cpu().numpy()      # 192 B per rank
    A_all = engine.g1_sum(np.ascontiguousarray(parts[:, :64]))
    B_all = engine.g2_sum(np.ascontiguousarray(parts[:, 64:]))
    A_all = A_all.cpu().numpy() if isinstance(A_all, torch.Tensor) else A_all
    B_all = B_all.cpu().numpy() if isinstance(B_all, torch.Tensor) else B_all
    P = np.concatenate([np.asarray(A_all).reshape(-1), np.asarray(g1).reshape(-1)])
    Q = np.concatenate([np.asarray(H).reshape(-1), neg(np.asarray(B_all).reshape(-1))])
    return bool(engine.pairing_check(P, Q))
