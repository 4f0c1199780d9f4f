"""GPU parity on STRUCTURED inputs (run with -m gpu): the places where a fast group law has a special case and random inputs never go.
  * scalars around the endomorphism splits of csrc/curve29.hip.hpp (GLV for G1: the eigenvalue lambda and its neighbours, halves that
    vanish or coincide; four-dimensional GLS for G2: powers and sums of mu = 6u^2), around r and around the window boundaries,
    through every size-dependent form of the kernels (one point per octet / quad of lanes / per lane);
  * point sums, fixed-base sums and bucket sums whose terms REPEAT or cancel (an addition that must double, or give infinity).
Checked against the C restatement and, on a subset, against the big-integer oracle (both under oracle/, test infrastructure).
Reference call sites: signature/bls01_signature/bls_signature.go:45,63 (ScalarMultiplication), bibe/afp25_bibe/afp25_bibe_utils.go:48,51,
cpabe/bsw07/bsw07_cpabe.go:149-160 (sums of multiples with attribute-dependent, possibly equal, bases)."""
import numpy as np
import pytest

import bn254_py as o

pytestmark = pytest.mark.gpu

U = 4965661367192848881                      # the BN parameter


@pytest.fixture(scope="module")
def eng():
    from gopairingbasedcryptography_amd import _build, bn254
    _build.build_library()
    bn254.init(0)
    return bn254


def _kb(ks):
    return np.frombuffer(b"".join(o.scalar_to_bytes(k) for k in ks), dtype=np.uint8)


def _cube_roots_of_unity():
    """lambda, lambda^2 (mod r): the two eigenvalues an order-3 endomorphism can have on a group of order r"""
    for g in range(2, 50):
        lam = pow(g, (o.R - 1) // 3, o.R)
        if lam != 1:
            assert (lam * lam + lam + 1) % o.R == 0
            return lam, lam * lam % o.R
    raise AssertionError("no generator found")


def structured_scalars():
    r = o.R
    lam, lam2 = _cube_roots_of_unity()
    mu = 6 * U * U % r
    ks = [0, 1, 2, 3, 4, 5, 7, 8, 15, 16, 17, r - 1, r - 2, r - 3, r, r + 1, r + 2, 2 * r, 2 * r + 1, 5 * r + 3, (1 << 256) - 1, (1 << 256) - 2, 1 << 255, (1 << 255) + 1]
    for e in (lam, lam2):                                             # one GLV half zero, halves equal, halves opposite, tiny halves
        ks += [e, e + 1, e - 1, r - e, r - e + 1, 2 * e, 3 * e, (e + 1) * 3 % r, (3 * e + 3) % r, (2 * e + 1) % r, (e + 2) % r, (1 - e) % r,
               ((1 << 127) * (e + 1)) % r, ((1 << 128) - 1) * (e + 1) % r, ((1 << 64) + (1 << 64) * e) % r, (e * e) % r, (e * e + e) % r]
    for i in range(1, 4):                                             # GLS: single sub-scalars, equal sub-scalars, alternating signs
        m = pow(mu, i, r)
        ks += [m, m + 1, m - 1, r - m, 2 * m % r, 3 * m % r, ((1 << 63) * m) % r, ((1 << 64) - 1) * m % r]
    ks += [(1 + mu + mu * mu + mu ** 3) % r, (1 - mu + mu * mu - mu ** 3) % r, (5 + 5 * mu + 5 * mu * mu + 5 * mu ** 3) % r,
           ((1 << 65) * (1 + mu)) % r, (3 + 2 * mu + mu * mu) % r, U, U + 1, 6 * U + 2, 6 * U * U, 36 * U ** 3 % r, 36 * U ** 4 % r]
    for b in (1, 2, 29, 31, 32, 33, 58, 63, 64, 65, 96, 127, 128, 129, 130, 131, 159, 160, 191, 192, 224, 252, 253, 254):   # window and limb boundaries
        ks += [1 << b, (1 << b) - 1, (1 << b) + 1, r - (1 << b) if (1 << b) < r else (1 << b) - r]
    ks += [int("aa" * 32, 16), int("55" * 32, 16), int("33" * 32, 16), int("0f" * 32, 16), int("01" * 32, 16), int("ff" * 16, 16), int("ff" * 16 + "00" * 16, 16)]
    return [k % (1 << 256) for k in ks]


def test_scalar_mul_structured_scalars_every_form(eng, oracle):
    """[k]P for the scalars above, on random bases, the generator and infinity: the same (base, scalar) pairs through a call of 257
    (G2: one point per octet of lanes, G1: per quad), 4 099 (per quad) and 16 400 elements (per lane; the list repeated), and with ONE
    shared base (the quad kernel's shared-base form; from 16 384 scalars on the transient fixed-base table); against the C restatement
    for every pair and the big-integer oracle for a third of them."""
    ks = structured_scalars()
    m = len(ks)
    assert m > 150
    g1, g2 = eng.generators()
    seeds = _kb([o.bench_scalar("edgeP", i) for i in range(m)])
    for gen, mul, omul, w, pymul, frm, to in ((g1, eng.g1_scalar_mul, oracle.g1_scalar_mul, 64, o.g1_mul, o.g1_from_bytes, o.g1_to_bytes),
                                              (g2, eng.g2_scalar_mul, oracle.g2_scalar_mul, 128, o.g2_mul, o.g2_from_bytes, o.g2_to_bytes)):
        base = np.asarray(mul(gen, seeds)).reshape(m, w).copy()
        base[::7] = np.asarray(gen).reshape(1, w)                     # the generator among them
        base[5::31] = 0                                               # and infinity
        kb = _kb(ks).reshape(m, 32)
        want = np.asarray(omul(base.reshape(-1), kb.reshape(-1), threads=16)).reshape(m, w)
        for i in range(0, m, 3):                                      # the independent big-integer check
            assert want[i].tobytes() == to(pymul(frm(base[i].tobytes()), ks[i] % o.R) if base[i].any() else None), (w, i, hex(ks[i]))
        for n in (m, 4099, 16400):
            reps = -(-n // m)
            B = np.tile(base, (reps, 1))[:n]
            K = np.tile(kb, (reps, 1))[:n]
            got = np.asarray(mul(B.reshape(-1), K.reshape(-1))).reshape(n, w)
            bad = np.nonzero((got != np.tile(want, (reps, 1))[:n]).any(axis=1))[0]
            assert bad.size == 0, (w, n, bad[:8], [hex(ks[b % m]) for b in bad[:8]])
        # one shared base, every scalar (small call), and again from 16 384 scalars on (the table form)
        shared = np.asarray(omul(np.tile(base[1], (m, 1)).reshape(-1), kb.reshape(-1), threads=16)).reshape(m, w)
        assert (np.asarray(mul(base[1], kb.reshape(-1))).reshape(m, w) == shared).all(), w
        reps = -(-16400 // m)
        big = np.asarray(mul(base[1], np.tile(kb, (reps, 1))[:16400].reshape(-1))).reshape(16400, w)
        assert (big == np.tile(shared, (reps, 1))[:16400]).all(), w
        # ScalarMultiplicationBase through the generators' tables
        tab = np.asarray((eng.g1_scalar_mul_base if w == 64 else eng.g2_scalar_mul_base)(kb.reshape(-1))).reshape(m, w)      # the bytes as they are (values >= r included)
        assert (tab == np.asarray(omul(gen, kb.reshape(-1), threads=16)).reshape(m, w)).all(), w


def _neg(eng, pt, w):
    return np.frombuffer(o.g1_to_bytes(o.g1_neg(o.g1_from_bytes(pt.tobytes()))) if w == 64 else o.g2_to_bytes(o.g2_neg(o.g2_from_bytes(pt.tobytes()))), dtype=np.uint8)


def test_point_sums_of_repeated_and_opposite_points(eng, oracle):
    """Sums whose additions meet EQUAL points (must double) and OPPOSITE points (must give infinity) at every level of the tree:
    m copies of P equal [m]P; P and -P in any arrangement cancel; infinity anywhere changes nothing."""
    g1, g2 = eng.generators()
    for gen, mul, summ, osum, w in ((g1, eng.g1_scalar_mul, eng.g1_sum, oracle.g1_sum, 64), (g2, eng.g2_scalar_mul, eng.g2_sum, oracle.g2_sum, 128)):
        P = np.asarray(mul(gen, _kb([o.bench_scalar("rep", 0)]))).reshape(w)
        Q = np.asarray(mul(gen, _kb([o.bench_scalar("rep", 1)]))).reshape(w)
        nP, inf = _neg(eng, P, w), np.zeros(w, dtype=np.uint8)
        for m in (2, 3, 4, 5, 31, 32, 33, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4097):
            got = np.asarray(summ(np.tile(P, m))).reshape(w)
            assert (got == np.asarray(mul(P, _kb([m]))).reshape(w)).all(), (w, m, "copies")
            if m <= 257:
                assert (got == np.asarray(osum(np.tile(P, m))).reshape(w)).all(), (w, m, "oracle")
        rng = np.random.default_rng(99)
        for trial in range(12):
            n = int(rng.integers(2, 300))
            pick = rng.integers(0, 5, size=n)                         # 0: P, 1: -P, 2: infinity, 3: Q, 4: P again
            rows = np.stack([(P, nP, inf, Q, P)[i] for i in pick])
            count_p = int((pick == 0).sum() + (pick == 4).sum() - (pick == 1).sum())
            count_q = int((pick == 3).sum())
            want = np.asarray(summ(np.concatenate([np.asarray(mul(P, _kb([count_p % o.R]))).reshape(w), np.asarray(mul(Q, _kb([count_q]))).reshape(w)]))).reshape(w)
            got = np.asarray(summ(rows.reshape(-1))).reshape(w)
            assert (got == want).all(), (w, trial, n)
            assert (got == np.asarray(osum(rows.reshape(-1))).reshape(w)).all(), (w, trial, n, "oracle")
        assert not np.asarray(summ(np.concatenate([P, nP]))).any() and not np.asarray(summ(np.concatenate([np.tile(P, 8), np.tile(nP, 8)]))).any()
        assert not np.asarray(summ(np.concatenate([np.tile(np.concatenate([P, nP]), 100), inf]))).any()


def test_fixed_base_and_bucket_sums_with_repeated_bases(eng, oracle):
    """Sums of multiples whose TERMS coincide or cancel: the same base listed twice with equal scalars (the accumulator meets its own
    addend), with opposite scalars (infinity mid-sum), the negated base with the same scalar, bases at infinity, all scalars zero —
    through the fixed-base tables (few bases) and the bucket method (>= 16 384 terms)."""
    import torch
    g1, g2 = eng.generators()
    for gen, mul, summ, msm, w, is_g2 in ((g1, eng.g1_scalar_mul, eng.g1_sum, eng.g1_scalar_mul_sum, 64, False), (g2, eng.g2_scalar_mul, eng.g2_sum, eng.g2_scalar_mul_sum, 128, True)):
        A = np.asarray(mul(gen, _kb([o.bench_scalar("fbr", 0)]))).reshape(w)
        B = np.asarray(mul(gen, _kb([o.bench_scalar("fbr", 1)]))).reshape(w)
        nA, inf = _neg(eng, A, w), np.zeros(w, dtype=np.uint8)
        bases = np.stack([A, A, nA, inf, B, B, A, nA])
        a, b = o.bench_scalar("fbr", 2) % o.R, o.bench_scalar("fbr", 3) % o.R
        rows = [[a, a, a, a, b, (o.R - b) % o.R, 0, 0],               # A + A - A, B - B
                [a, (o.R - a) % o.R, 0, 5, b, b, a, a],               # A - A (infinity mid-sum), then B + B, A - A
                [1, 1, 1, 1, 1, 1, 1, 1],
                [0, 0, 0, 0, 0, 0, 0, 0],
                [255, 255, 255, 1, 256, 256, 65535, 65535],           # single-window digits: table rows of equal index
                [o.R - 1, o.R - 1, o.R - 1, 7, 2, o.R - 2, 1, 1]]
        fb = eng.FixedBase(bases.reshape(-1), g2=is_g2)
        got = np.asarray(fb.msm(np.stack([_kb(r) for r in rows]))).reshape(len(rows), w)
        for i, r in enumerate(rows):
            want = np.asarray(summ(mul(bases.reshape(-1), _kb(r)))).reshape(w)
            assert (got[i] == want).all(), (w, i, "fixed base")
            assert (np.asarray(msm(bases.reshape(-1), _kb(r))).reshape(w) == want).all(), (w, i, "small sum of multiples")
        fb.close()
        # the bucket method: 16 500 terms over FOUR distinct points and their negatives, scalars drawn from a handful of values, so that
        # buckets hold runs of equal and of opposite points (each bucket's running sum meets doublings and infinities)
        rng = np.random.default_rng(2024)
        n = 16500
        pool_pts = np.stack([A, nA, B, _neg(eng, B, w), inf])
        pool_k = _kb([a, b, (o.R - a) % o.R, 1, 2, (1 << 255) - 19, 12345678901234567890, 0]).reshape(8, 32)
        pts = pool_pts[rng.integers(0, 5, size=n)]
        ks = pool_k[rng.integers(0, 8, size=n)]
        dp, dk = torch.from_numpy(np.ascontiguousarray(pts)).cuda(), torch.from_numpy(np.ascontiguousarray(ks)).cuda()
        want = summ(mul(dp.reshape(-1), dk.reshape(-1))).cpu().numpy().reshape(w)
        assert (msm(dp.reshape(-1), dk.reshape(-1)).cpu().numpy().reshape(w) == want).all(), (w, "bucket / per-term path")
        # ... and against the restatement on a 600-term slice
        sl = slice(0, 600)
        o_terms = (oracle.g2_scalar_mul if is_g2 else oracle.g1_scalar_mul)(pts[sl].reshape(-1), ks[sl].reshape(-1), threads=16)
        o_want = np.asarray((oracle.g2_sum if is_g2 else oracle.g1_sum)(np.asarray(o_terms).reshape(-1))).reshape(w)
        assert (np.asarray(msm(pts[sl].reshape(-1), ks[sl].reshape(-1))).reshape(w) == o_want).all(), (w, "oracle slice")


def test_pairing_of_repeated_and_related_points(eng, oracle):
    """Pairings whose inputs are related: the same pair many times (every lane of a wavefront on one path), P and -P against one Q,
    Q and -Q, multiples of the generators by tiny scalars (the G2 walk meets small-order arithmetic nowhere, but the line values are
    as non-random as they get), in the latency form, the pipelined form and the throughput kernels."""
    g1, g2 = eng.generators()
    small = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, o.R - 1, o.R - 2, U, 6 * U + 2, 6 * U * U % o.R]
    P = np.asarray(eng.g1_scalar_mul(g1, _kb(small))).reshape(len(small), 64)
    Q = np.asarray(eng.g2_scalar_mul(g2, _kb(small))).reshape(len(small), 128)
    pairs_p, pairs_q = [], []
    for i in range(len(small)):
        for j in range(len(small)):
            pairs_p.append(P[i]); pairs_q.append(Q[j])
    Pm, Qm = np.stack(pairs_p), np.stack(pairs_q)
    n = Pm.shape[0]
    want = np.asarray(oracle.pair_batch(Pm.reshape(-1), Qm.reshape(-1), threads=16)).reshape(n, 384)
    got = np.asarray(eng.pair_batch(Pm.reshape(-1), Qm.reshape(-1))).reshape(n, 384)      # <= 2 048 pairs: the latency form
    assert (got == want).all()
    from gopairingbasedcryptography_amd import _lib
    _lib.check(_lib.load().gpbc_set_latency_path(0))
    try:
        assert (np.asarray(eng.pair_batch(Pm.reshape(-1), Qm.reshape(-1))).reshape(n, 384) == want).all()       # pipelined form
        reps = 40                                                     # > 16 384 pairs: the two throughput kernels
        big = np.asarray(eng.pair_batch(np.tile(Pm, (reps, 1)).reshape(-1), np.tile(Qm, (reps, 1)).reshape(-1))).reshape(reps * n, 384)
        assert (big == np.tile(want, (reps, 1))).all()
    finally:
        _lib.load().gpbc_set_latency_path(2048)
    # e([a]g1, [b]g2) depends on a b only: rows with equal products of small scalars are equal
    idx = {}
    for i, a in enumerate(small[:16]):
        for j, b in enumerate(small[:16]):
            idx.setdefault(a * b, []).append(i * len(small) + j)
    for rows in idx.values():
        assert all((got[r] == got[rows[0]]).all() for r in rows)
    # PairingCheck of cancelling segments: e(P, Q) e(-P, Q) and e(P, Q) e(P, -Q), many times in one call
    nP = np.stack([_neg(eng, P[i], 64) for i in range(8)])
    nQ = np.stack([_neg(eng, Q[i], 128) for i in range(8)])
    segP = np.concatenate([np.stack([P[i], nP[i]]) for i in range(8)] + [np.stack([P[i], P[i]]) for i in range(8)])
    segQ = np.concatenate([np.stack([Q[i], Q[i]]) for i in range(8)] + [np.stack([Q[i], nQ[i]]) for i in range(8)])
    seg = np.arange(0, 33, 2, dtype=np.uint64)
    gt = np.asarray(eng.multi_pair(segP.reshape(-1), segQ.reshape(-1), seg)).reshape(16, 384)
    one = np.frombuffer(o.gt_to_bytes(o.F12_ONE), dtype=np.uint8)
    assert all((gt[i] == one).all() for i in range(16))
