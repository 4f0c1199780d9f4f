"""Every GPU configuration of BASELINE.json at its STATED size on one MI355X (run with -m gpu), through the C ABI:

  configs[1]  2^20 independent bn254.Pair                          oracle spot checks around every 262144-pair workspace boundary
                                                                    + identities over the whole batch
  configs[2]  BLS aggregate verification of 2^20 signatures         partial sums vs the oracle on a 4096 prefix, accept / forged-reject
  configs[3]  BSW07 decrypt, 256-attribute policy, 2^16 ciphertexts 256-of-256 and 16 x (16-of-16): every message recovered,
                                                                    oracle reference-shaped evaluation on 4 ciphertexts
  configs[4]  AFP25 batch decryption of 2^18 identities, B = 256    every message recovered, host planner (digest, quotients,
                                                                    fixed-base openings) on whole batches, oracle on 4 items

Inputs are the synthetic workloads of SURVEY.md §8d (bench_workloads.py, the same generators bench.py times).  The oracle
is the checker only; sizes it cannot finish in seconds are covered by the schemes' own round trips (decrypt returns the
encrypted message for EVERY ciphertext) and by algebraic identities over the full batch.
"""
import numpy as np
import pytest

import bn254_py as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from gopairingbasedcryptography_amd import _build, bn254
    _build.build_library()
    bn254.init(0)
    return bn254


@pytest.fixture(scope="module")
def dev():
    import torch
    return torch.device("cuda", 0)


def test_config1_pairs_2_20(eng, oracle, dev):
    """2^20 pairs on the bench's own points P_i = [k("P",i)] g1, Q_i = [k("Q",i)] g2 (four 262144-pair workspace chunks)."""
    import torch
    import bench_workloads as w
    n = 1 << 20
    g1, g2 = eng.generators()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    P = eng.g1_scalar_mul(d(g1), d(w.bench_scalars("P", 0, n)))
    Q = eng.g2_scalar_mul(d(g2), d(w.bench_scalars("Q", 0, n)))
    gt = eng.pair_batch(P, Q)
    torch.cuda.synchronize()
    edges = [0, 1, 63, 64, 65]
    for c in (1, 2, 3):
        edges += [c * 262144 - 2, c * 262144 - 1, c * 262144, c * 262144 + 1, c * 262144 + 63, c * 262144 + 64]
    edges += [n - 65, n - 64, n - 2, n - 1, 131071, 500000, 777777, 1000003]
    idx = np.array(edges)
    Ps, Qs = P[idx].cpu().numpy(), Q[idx].cpu().numpy()
    assert (Ps == oracle.g1_scalar_mul(g1, np.concatenate([w.bench_scalars("P", int(i), 1) for i in idx]), threads=8)).all()
    assert (gt[idx].cpu().numpy() == oracle.pair_batch(Ps, Qs, threads=16)).all()
    # identities over all 2^20 outputs: e([3]P, Q) == e(P,Q)^2 * e(P,Q) and e(P, Q) * e(P, -Q)... via the second batch
    three = d(np.tile(np.frombuffer((3).to_bytes(32, "little"), dtype=np.uint8), n).reshape(n, 32).copy())
    gt3 = eng.pair_batch(eng.g1_scalar_mul(P, three), Q)
    assert bool((gt3 == eng.gt_mul(eng.gt_mul(gt, gt), gt)).all())
    # every output is in the order-r subgroup's image under inversion: gt * gt^-1 == 1 and no output is one
    one = np.frombuffer(o.gt_to_bytes(o.F12_ONE), dtype=np.uint8)
    prod = eng.gt_mul(gt, eng.gt_inverse(gt))
    assert bool((prod == d(one)).all()) and not bool((gt == d(one)).all(dim=1).any())
    # GT.Exp over the whole batch (pairing values: the Granger-Scott path of k_gt_exp): x^a x^b == x^(a+b) with two 254-bit exponents
    # per element, oracle on a few of them; and the same identity on 2^16 Miller values (outside the cyclotomic subgroup: general path)
    ka, kb = w.bench_scalars("ea", 0, n).reshape(n, 32), w.bench_scalars("eb", 0, n).reshape(n, 32)
    la, lb = ka.copy().view("<u8").astype(object), kb.copy().view("<u8").astype(object)
    carry, limbs = np.zeros(n, dtype=object), []
    for j in range(4):
        t = la[:, j] + lb[:, j] + carry
        limbs.append((t & 0xFFFFFFFFFFFFFFFF).astype(np.uint64)); carry = t >> 64
    ksum = np.ascontiguousarray(np.stack(limbs, axis=1)).view(np.uint8).reshape(n, 32)
    xa, xb, xs = eng.gt_exp(gt, d(ka)), eng.gt_exp(gt, d(kb)), eng.gt_exp(gt, d(ksum))
    assert bool((eng.gt_mul(xa, xb) == xs).all())
    pick = np.array([0, 31, 32, 4097, n - 1])
    assert (xs[pick].cpu().numpy() == oracle.gt_exp(gt[pick].cpu().numpy(), ksum[pick].reshape(-1), threads=5)).all()
    m = 1 << 16
    f = eng.miller_loop(P[:m].contiguous(), Q[:m].contiguous())
    fa, fb, fs = eng.gt_exp(f, d(ka[:m])), eng.gt_exp(f, d(kb[:m])), eng.gt_exp(f, d(ksum[:m]))
    assert bool((eng.gt_mul(fa, fb) == fs).all())
    assert (fs[pick[:4]].cpu().numpy() == oracle.gt_exp(f[pick[:4]].cpu().numpy(), ksum[pick[:4]].reshape(-1), threads=4)).all()


def test_config2_aggregate_verify_2_20(eng, oracle, dev):
    import bench_workloads as w
    n = 1 << 20
    inst = w.aggregate(eng, n, dev)
    A = eng.g1_scalar_mul_sum(inst["pk"], inst["rho"]).cpu().numpy()
    B = eng.g2_scalar_mul_sum(inst["sigma"], inst["rho"]).cpu().numpy()
    # the G2 sum equals [sum rho_i x_i] H (what the synthetic harness may substitute when only the G1 side is timed)
    assert (B == eng.g2_scalar_mul(inst["H"], [inst["sum_rho_x"]])[0]).all()
    assert w.aggregate_check(eng, A, B, inst["H"], inst["g1"])
    # partial sums against the oracle on the first 4096 signatures
    m = 4096
    pk, sg, rho = inst["pk"][:m].contiguous(), inst["sigma"][:m].contiguous(), inst["rho"][:m].contiguous()
    Ap, Bp = eng.g1_scalar_mul_sum(pk, rho).cpu().numpy(), eng.g2_scalar_mul_sum(sg, rho).cpu().numpy()
    assert (Ap == oracle.g1_sum(oracle.g1_scalar_mul(pk.cpu().numpy(), rho.cpu().numpy(), threads=16))).all()
    assert (Bp == oracle.g2_sum(oracle.g2_scalar_mul(sg.cpu().numpy(), rho.cpu().numpy(), threads=16))).all()
    # one forged signature among 2^20 is rejected
    forged = inst["sigma"].clone()
    forged[n // 3] = forged[n // 3 + 1]
    Bf = eng.g2_scalar_mul_sum(forged, inst["rho"]).cpu().numpy()
    assert not w.aggregate_check(eng, A, Bf, inst["H"], inst["g1"])
    # the host-pointer entry (sharded over the bound devices) gives the same sums
    h = 1 << 16
    assert (eng.g1_scalar_mul_sum(inst["pk"][:h].cpu().numpy(), inst["rho"][:h].cpu().numpy()) ==
            eng.g1_scalar_mul_sum(inst["pk"][:h].contiguous(), inst["rho"][:h].contiguous()).cpu().numpy()).all()


@pytest.mark.parametrize("kind", ["256of256", "16x16"])
def test_config3_bsw07_2_16_ciphertexts(eng, oracle, dev, kind):
    """cpabe/bsw07 Decrypt with a 256-attribute policy on 2^16 ciphertexts (33.6 M Miller loops, 2^16 final exponentiations)
    through bsw07.decrypt_plan / fold_key / decrypt_batch_arrays."""
    import bench_workloads as w
    from bsw07_fixture import Instance
    from gopairingbasedcryptography_amd import bsw07
    n = 1 << 16
    inst = w.bsw07_instance(eng, kind, n, dev)
    assert inst["pairs_per_ct"] == 513
    plan = bsw07.decrypt_plan(inst["tree"], inst["attrs"])
    assert len(plan) == 256
    folded = bsw07.fold_key(eng, plan, inst["dj"], inst["dj_prime"])
    out = bsw07.decrypt_batch_arrays(eng, folded, inst["D"], inst["c_tilde"], inst["c"], inst["cy"], inst["cy_prime"])
    assert bool((out == inst["msgs"]).all())                           # every one of the 2^16 messages comes back
    # the dict-form planner on a few ciphertexts, and the reference-shaped evaluation (pairing by pairing, GT.Div / Exp / Mul as
    # access/tree/access_tree_node.go:96-164 and bsw07_cpabe.go:172-195) by the oracle on 4 of them
    pick = [0, 1, n // 2, n - 1]
    cts = [w.bsw07_ct_dict(inst, j) for j in pick]
    small = bsw07.decrypt_batch(eng, folded, inst["D"], cts, eng.g1_neg)
    ref = Instance.__new__(Instance)
    ref.tree, ref.user_attrs, ref.dj, ref.dj_prime, ref.D = inst["tree"], inst["attrs"], inst["dj"], inst["dj_prime"], inst["D"]
    for t, j in enumerate(pick):
        want = inst["msgs"][j].cpu().numpy()
        assert (small[t] == want).all()
        assert (ref.reference_shaped_decrypt(oracle, cts[t]) == want).all()
    # a key without attribute 7 does not satisfy an n-of-n policy
    assert bsw07.decrypt_plan(inst["tree"], inst["attrs"] - {7}) is None


def test_config4_afp25_2_18_identities(eng, oracle, dev):
    """bibe/afp25_bibe batch decryption of 2^18 (ciphertext, identity) items in batches of B = 256."""
    import bench_workloads as w
    from gopairingbasedcryptography_amd import afp25
    n, B = 1 << 18, 256
    inst = w.afp25_instance(eng, B, n, dev)
    out = afp25.decrypt_batch_arrays(eng, inst["D"], inst["pi"], inst["sk"], inst["C1"], inst["C2"])
    assert bool((out == inst["msgs"]).all())                           # all 2^18 messages
    # the host planner on two whole batches: digest from the SRS, quotients by synthetic division, openings by ONE fixed-base
    # MSM call per batch — the same D and pi the generator derived from tau, and the same messages through decrypt_batch
    srs = w.afp25_srs(eng, inst)
    table = afp25.srs_table(eng, inst["g1"], srs)
    for b in (0, n // B - 1):
        ids = inst["ids"][b * B:(b + 1) * B]
        D, f = afp25.digest(eng, inst["g1"], srs, ids)
        assert (np.asarray(D) == inst["D"][b * B].cpu().numpy()).all()
        pis = afp25.commit_g1_many(table, [afp25.quotient_by_root(f, i) for i in ids])
        assert (pis == inst["pi"][b * B:(b + 1) * B].cpu().numpy()).all()
        items = [(ids[t], inst["C1"][b * B + t].cpu().numpy(), inst["C2"][b * B + t].cpu().numpy()) for t in range(0, B, 37)]
        got = afp25.decrypt_batch(eng, inst["g1"], srs, D, f, inst["sk"][b * B].cpu().numpy(), items, table=table, identities=ids)
        assert (got == inst["msgs"][b * B:(b + 1) * B:37].cpu().numpy()).all()
    table.close()
    # reference-shaped evaluation by the oracle on 4 items: three pairings, two GT.Mul, one GT.Div (afp25_bibe.go:395-413)
    for j in (0, 255, n // 2 + 1, n - 1):
        C1, C2 = inst["C1"][j].cpu().numpy(), inst["C2"][j].cpu().numpy()
        p = [oracle.pair_batch(inst[k][j].cpu().numpy(), C1[i]) for i, k in enumerate(("D", "pi", "sk"))]
        want = oracle.gt_div(C2, oracle.gt_mul(oracle.gt_mul(p[0], p[1]), p[2]))[0]
        assert (want == out[j].cpu().numpy()).all() and (want == inst["msgs"][j].cpu().numpy()).all()


def test_pipelined_host_entries_match_the_device_path(eng, dev):
    """Host-pointer calls of 2 x 131072 units or more run as chunks alternating on two streams (upload / kernels / download of
    neighbouring chunks overlap: csrc/gpbc_common.hpp pipelined_chunks).  300 001 units = two full chunks and a ragged third;
    the results must be the bytes of the HBM-resident path."""
    import torch
    import bench_workloads as w
    n = 300001
    g1, g2 = eng.generators()
    d = lambda a: torch.from_numpy(np.array(a, dtype=np.uint8, copy=True)).to(dev)
    kP, kQ, ks = (d(w.bench_scalars(t, 7, n)).reshape(n, 32) for t in ("P", "Q", "s"))
    P, Q = eng.g1_scalar_mul(d(g1), kP), eng.g2_scalar_mul(d(g2), kQ)
    gt = eng.pair_batch(P, Q)
    Ph, Qh, kh = P.cpu().numpy(), Q.cpu().numpy(), ks.cpu().numpy()
    assert (eng.pair_batch(Ph, Qh) == gt.cpu().numpy()).all()
    assert (eng.g1_scalar_mul(Ph, kh) == eng.g1_scalar_mul(P, ks).cpu().numpy()).all()
    assert (eng.g2_scalar_mul(Qh, kh) == eng.g2_scalar_mul(Q, ks).cpu().numpy()).all()


def test_beyond_the_baseline_size(eng, oracle, dev):
    """Four times BASELINE's batch and ragged: 2^22 + 17 pairs in ONE device call (seventeen 262 144-pair workspace chunks, the last of
    17 pairs) and as many scalar multiplications in each group; a window that straddles the 2^22 boundary and windows elsewhere must be
    the bytes of separate small calls, the tail and a few interior rows the oracle's.  (Sizes are size_t throughout the ABI; this is the
    largest batch the suite runs — 2.4 GB of points and results in HBM.)"""
    import torch
    import bench_workloads as w
    n = (1 << 22) + 17
    g1, g2 = eng.generators()
    d = lambda a: torch.from_numpy(np.array(a, dtype=np.uint8, copy=True)).to(dev)
    rng = np.random.default_rng(4222)
    kP, kQ = rng.integers(0, 256, size=(n, 32), dtype=np.uint8), rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    kP[:, 31] &= 0x1F; kQ[:, 31] &= 0x1F                                          # below 2^253 < r
    kP, kQ = d(kP), d(kQ)
    P, Q = eng.g1_scalar_mul(d(g1), kP), eng.g2_scalar_mul(d(g2), kQ)             # one shared base: the transient fixed-base table
    assert P.shape[0] == n and Q.shape[0] == n
    gt = eng.pair_batch(P, Q)
    R1, R2 = eng.g1_scalar_mul(P, kQ), eng.g2_scalar_mul(Q, kP)                   # one base per scalar: the variable-base kernels
    torch.cuda.synchronize()
    for lo, hi in ((0, 300), (262144 - 100, 262144 + 100), ((1 << 22) - 200, n), (3 * (1 << 20) + 12345, 3 * (1 << 20) + 12345 + 2500)):
        assert torch.equal(eng.pair_batch(P[lo:hi].contiguous(), Q[lo:hi].contiguous()), gt[lo:hi]), (lo, hi)
        assert torch.equal(eng.g1_scalar_mul(P[lo:hi].contiguous(), kQ[lo:hi].contiguous()), R1[lo:hi]), (lo, hi)
        assert torch.equal(eng.g2_scalar_mul(Q[lo:hi].contiguous(), kP[lo:hi].contiguous()), R2[lo:hi]), (lo, hi)
    idx = np.array([0, 1, (1 << 21) + 3, (1 << 22) - 1, 1 << 22, n - 2, n - 1])
    Ps, Qs, k1, k2 = P[idx].cpu().numpy(), Q[idx].cpu().numpy(), kQ[idx].cpu().numpy(), kP[idx].cpu().numpy()
    assert (gt[idx].cpu().numpy() == oracle.pair_batch(Ps, Qs, threads=8)).all()
    assert (R1[idx].cpu().numpy() == oracle.g1_scalar_mul(Ps, k1.reshape(-1), threads=8)).all()
    assert (R2[idx].cpu().numpy() == oracle.g2_scalar_mul(Qs, k2.reshape(-1), threads=8)).all()
    one = torch.from_numpy(np.frombuffer(o.gt_to_bytes(o.F12_ONE), dtype=np.uint8).copy()).to(dev)
    assert not bool((gt == one).all(dim=1).any())                                 # no output is one, none is zero
    assert bool(gt.any(dim=1).all())
