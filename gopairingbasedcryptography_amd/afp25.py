"""Host-side planner for batched AFP25 batched-IBE decryption (BASELINE config 5).

Reference flow (bibe/afp25_bibe/afp25_bibe.go:369-418, afp25_bibe_utils.go:14-55): to decrypt for identity id inside a
batch of B identities, rebuild the quotient polynomial q(X) = f(X)/(X - id) by multiplying out B-1 linear factors
(O(B^2) Fr operations), commit to it with B separate G1 scalar multiplications and B affine additions, run THREE full
pairings e(D, C1[0]), e(pi, C1[1]), e(sk, C1[2]), multiply them and divide C2 by the product.

Batched form used here, bit-identical in its GT result:
  * f(X) is expanded once per batch; every q_i(X) comes from it by synthetic division (O(B) each);
  * pi_i = sum_j coef_j * [tau^j]_1 through the engine's G1 scalar-multiplication kernel and point-sum tree;
  * all items of the batch go through ONE multi_pair call (3 Miller loops + 1 final exponentiation per item) and one gt_div.
Host orchestration only; all group arithmetic goes through the engine (`bn254`).
"""
import numpy as np

R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def poly_from_roots(roots):
    """Coefficients (constant term first) of prod (X - root) mod r  (computePolynomialCoeffs, afp25_bibe_utils.go:14-43)."""
    coeffs = [1]
    for root in roots:
        new = [0] * (len(coeffs) + 1)
        for i, c in enumerate(coeffs):
            new[i] = (new[i] - root * c) % R_ORDER
            new[i + 1] = (new[i + 1] + c) % R_ORDER
        coeffs = new
    return coeffs


def quotient_by_root(coeffs, root):
    """f(X) / (X - root) by synthetic division; raises if root is not a root (identity not in the batch)."""
    n = len(coeffs) - 1
    q = [0] * n
    carry = 0
    for i in range(n, 0, -1):
        carry = (coeffs[i] + carry * root) % R_ORDER
        q[i - 1] = carry
    if (coeffs[0] + carry * root) % R_ORDER != 0:
        raise ValueError("identity not found in identity list")
    return q


def commit_g1(engine, g1, tau_powers, coeffs):
    """[p(tau)]_1 = coef_0*g1 + sum_j coef_j*[tau^j]_1  (computeG1PolynomialTau, afp25_bibe_utils.go:45-55)."""
    pts = np.concatenate([np.asarray(g1, dtype=np.uint8).reshape(1, 64), np.asarray(tau_powers, dtype=np.uint8)[: len(coeffs) - 1]])
    if hasattr(engine, "g1_scalar_mul_sum"):                    # one call: sum_j [c_j] srs_j (the bucket method from 16 384 terms on)
        return np.asarray(engine.g1_scalar_mul_sum(pts, [int(c) for c in coeffs]))
    return np.asarray(engine.g1_sum(engine.g1_scalar_mul(pts, [int(c) for c in coeffs])))


def digest(engine, g1, tau_powers, identities):
    """Batch digest D = [f(tau)]_1, f(X) = prod (X - id)  (Digest, afp25_bibe.go:293-305). Returns (D, f coefficients)."""
    if len(identities) == 0:
        raise ValueError("identities is empty")
    if len(identities) > len(tau_powers):
        raise ValueError("too many identities for batch size")
    f = poly_from_roots(identities)
    return commit_g1(engine, g1, tau_powers, f), f


def srs_table(engine, g1, tau_powers):
    """Fixed-base window tables (engine.FixedBase, HBM-resident) over the commitment bases (g1, [tau]_1, ..., [tau^B]_1):
    built once per SRS, after which every commitment is ONE row of a multi-scalar multiplication."""
    return engine.FixedBase(np.concatenate([np.asarray(g1, dtype=np.uint8).reshape(1, 64), np.asarray(tau_powers, dtype=np.uint8).reshape(-1, 64)]))


def commit_g1_many(table, coeff_rows):
    """[p_i(tau)]_1 for every coefficient row (lowest degree first; rows are zero-padded to the table's nbase)."""
    rows = [[int(c) for c in r] + [0] * (table.nbase - len(r)) for r in coeff_rows]
    return np.asarray(table.msm(rows))


def decrypt_batch(engine, g1, tau_powers, D, f_coeffs, sk, items, table=None, identities=None):
    """items: list of (identity, C1 [3,128], C2 [384]) all encrypted under the batch digest D and key sk.
    identities: the batch's identity list (what Digest was computed from); when given, the reference's error behaviour for
    absent and for duplicated identities is reproduced (a duplicated identity is still a root of f, so the quotient alone
    would not notice it).
    Returns the messages [n,384]: m_i = C2_i / (e(D, C1_i[0]) e(pi_i, C1_i[1]) e(sk, C1_i[2])).
    With `table` (srs_table) all opening proofs pi_i come from one fixed-base MSM call instead of one scalar-multiplication
    batch and one point sum per item."""
    if identities is not None:
        # the reference's Decrypt removes every identity equal to id from the batch list and fails unless exactly one was
        # removed (bibe/afp25_bibe/afp25_bibe.go:371-381): an identity that is absent OR duplicated in the batch is an error
        for ident, _, _ in items:
            if sum(1 for x in identities if x % R_ORDER == ident % R_ORDER) != 1:
                raise ValueError("identity not found in identity list")
    if table is not None:
        pis = commit_g1_many(table, [quotient_by_root(f_coeffs, ident) for ident, _, _ in items])
    else:
        pis = [commit_g1(engine, g1, tau_powers, quotient_by_root(f_coeffs, ident)) for ident, _, _ in items]
    P_rows, Q_rows, c2 = [], [], []
    for (ident, C1, C2), pi in zip(items, pis):
        P_rows.append(np.stack([np.asarray(D, dtype=np.uint8), np.asarray(pi, dtype=np.uint8).reshape(64), np.asarray(sk, dtype=np.uint8)]))
        Q_rows.append(np.asarray(C1, dtype=np.uint8).reshape(3, 128))
        c2.append(np.asarray(C2, dtype=np.uint8))
    off = np.arange(0, 3 * len(items) + 1, 3)
    X = engine.multi_pair(np.concatenate(P_rows), np.concatenate(Q_rows), off)
    return engine.gt_div(np.stack(c2), X)


def decrypt_batch_arrays(engine, D, pi, sk, C1, C2):
    """The pairing part of the batch decryption on arrays (numpy, or CUDA tensors): item i is decrypted with its batch's
    digest D[i], its opening proof pi[i] and key sk[i] — m_i = C2_i / (e(D_i, C1_i[0]) e(pi_i, C1_i[1]) e(sk_i, C1_i[2]))
    (bibe/afp25_bibe/afp25_bibe.go:395-413): ONE multi-pairing of 3-pair segments and one gt_div, BASELINE config 5 at its
    stated size.  D, pi, sk: [n,64]; C1: [n,3,128]; C2: [n,384]."""
    if type(D).__module__.startswith("torch"):
        import torch
        n = D.numel() // 64
        P = torch.stack([D.reshape(n, 64), pi.reshape(n, 64), sk.reshape(n, 64)], dim=1).contiguous()
        X = engine.multi_pair(P.reshape(-1), C1.reshape(-1).contiguous(), np.arange(0, 3 * n + 1, 3, dtype=np.uint64))
        return engine.gt_div(C2.reshape(n, 384).contiguous(), X)
    n = np.asarray(D).size // 64
    P = np.stack([np.asarray(D, dtype=np.uint8).reshape(n, 64), np.asarray(pi, dtype=np.uint8).reshape(n, 64),
                  np.asarray(sk, dtype=np.uint8).reshape(n, 64)], axis=1)
    X = engine.multi_pair(P, np.asarray(C1, dtype=np.uint8).reshape(3 * n, 128), np.arange(0, 3 * n + 1, 3))
    return engine.gt_div(np.asarray(C2, dtype=np.uint8).reshape(n, 384), X)
