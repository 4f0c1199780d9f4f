"""Host-side planner for batched BSW07 CP-ABE decryption (BASELINE config 4).

The reference decrypts one ciphertext at a time by walking the threshold access tree
(access/tree/access_tree_node.go:96-164, cpabe/bsw07/bsw07_cpabe.go:172-195): per satisfied leaf two full pairings and a
GT division, per threshold node a GT.Exp by a Lagrange coefficient and a GT.Mul, then e(C, D) and two more divisions.
Mathematically

    M = C~ * A / e(C, D),      A = prod_leaves ( e(Cy, Dj) / e(Cy', Dj') ) ^ Delta_leaf

where Delta_leaf is the product of the Lagrange coefficients on the path from the leaf to the root.  Because
e(P, Q)^k = e(P, [k]Q) and 1/e(P, Q) = e(P, -Q) hold exactly as Fp12 elements, the whole decryption is ONE
multi-pairing with ONE final exponentiation per ciphertext once the exponents are folded into the key:

    Dj^ = [Delta] Dj,   Dj'^ = [-Delta] Dj'          (2 G2 scalar multiplications per used leaf, once per key and policy)
    M   = C~ * Pair([Cy..., Cy'..., -C], [Dj^..., Dj'^..., D])      (per ciphertext: 2l+1 Miller loops, 1 final exp)

This module is host orchestration only (tree walking and Lagrange coefficients in Fr, microseconds per policy); every
group operation goes through the engine (`bn254`: g2_scalar_mul, multi_pair, gt_mul).  It does not reproduce the
reference's debug pairing / prints (access_tree_node.go:99-100,116-127).
"""
import numpy as np

R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class Leaf:
    """Leaf of the access tree: satisfied when `attribute` is in the user's attribute set."""
    def __init__(self, attribute):
        self.attribute = attribute
        self.leaf_id = None


class Threshold:
    """k-of-n gate; children are numbered 1..n as in NewThresholdNode (access_tree_node.go:38-52)."""
    def __init__(self, k, *children):
        if k < 1 or k > len(children):
            raise ValueError("threshold must be between 1 and len(children)")
        self.k = k
        self.children = list(children)


def assign_leaf_ids(node, start=1):
    """Depth-first leaf numbering from 1 (GenerateLeafID, access_tree_node.go:77-94). Returns the next free id."""
    if isinstance(node, Leaf):
        node.leaf_id = start
        return start + 1
    for c in node.children:
        start = assign_leaf_ids(c, start)
    return start


def lagrange_at_zero(index, indices):
    """Delta_{index,S}(0) = prod_{j in S, j != index} (0 - j)/(index - j) mod r  (utils/compute_lagrange_basis.go:8-30)."""
    num, den = 1, 1
    for j in indices:
        if j != index:
            num = num * (-j) % R_ORDER
            den = den * (index - j) % R_ORDER
    return num * pow(den, -1, R_ORDER) % R_ORDER


def decrypt_plan(node, attributes):
    """{leaf_id: (attribute, Delta)} for the leaves the reference's DecryptNode would use (the first `k` satisfied
    children of every gate, in child order), or None when the attribute set does not satisfy the tree."""
    if isinstance(node, Leaf):
        return {node.leaf_id: (node.attribute, 1)} if node.attribute in attributes else None
    chosen = []
    for i, c in enumerate(node.children, start=1):
        sub = decrypt_plan(c, attributes)
        if sub is not None:
            chosen.append((i, sub))
            if len(chosen) == node.k:
                break
    if len(chosen) < node.k:
        return None
    idx = [i for i, _ in chosen]
    plan = {}
    for i, sub in chosen:
        d = lagrange_at_zero(i, idx)
        for leaf_id, (attr, coef) in sub.items():
            plan[leaf_id] = (attr, coef * d % R_ORDER)
    return plan


def fold_key(engine, plan, dj, dj_prime):
    """Fold the Lagrange coefficients into the user's key: rows ordered by leaf id.
    dj, dj_prime: {attribute: 128-byte G2 point}.  Returns (leaf_ids, Dj^ [l,128], Dj'^ [l,128])."""
    leaf_ids = sorted(plan)
    base = np.stack([np.asarray(dj[plan[i][0]], dtype=np.uint8) for i in leaf_ids])
    base_p = np.stack([np.asarray(dj_prime[plan[i][0]], dtype=np.uint8) for i in leaf_ids])
    pos = [plan[i][1] for i in leaf_ids]
    neg = [(-c) % R_ORDER for c in pos]
    return leaf_ids, engine.g2_scalar_mul(base, pos), engine.g2_scalar_mul(base_p, neg)


def decrypt_batch(engine, folded, d_key, cts, neg_g1):
    """Decrypt n ciphertexts under one (key, policy) plan.

    folded = fold_key(...) result; d_key = D (128 B);  cts = list of dicts with 'c_tilde' (384 B), 'c' (64 B),
    'cy' / 'cy_prime' ({leaf_id: 64 B});  neg_g1 negates an affine G1 point on the host (gnark's G1Affine.Neg).
    Returns the n messages as an [n,384] array: one fixed-Q multi-pairing call (n segments of 2l+1 pairs against the one
    folded key) and one gt_mul call."""
    leaf_ids, dj_hat, djp_hat = folded
    l = len(leaf_ids)
    Q_seg = np.concatenate([np.asarray(dj_hat).reshape(l, 128), np.asarray(djp_hat).reshape(l, 128),
                            np.asarray(d_key, dtype=np.uint8).reshape(1, 128)])
    P_rows, ctil = [], []
    for ct in cts:
        P_rows.append(np.stack([np.asarray(ct["cy"][i], dtype=np.uint8) for i in leaf_ids]
                               + [np.asarray(ct["cy_prime"][i], dtype=np.uint8) for i in leaf_ids]
                               + [neg_g1(np.asarray(ct["c"], dtype=np.uint8))]))
        ctil.append(np.asarray(ct["c_tilde"], dtype=np.uint8))
    m = 2 * l + 1
    if hasattr(engine, "multi_pair_fixed_q"):
        # the folded key is the same G2 list for every ciphertext: its Miller lines are computed once for the whole batch
        X = engine.multi_pair_fixed_q(np.concatenate(P_rows), Q_seg)
    else:
        off = np.arange(0, len(cts) * m + 1, m)
        X = engine.multi_pair(np.concatenate(P_rows), np.concatenate([Q_seg] * len(cts)), off)
    return engine.gt_mul(np.stack(ctil), X)                                          # C~ * A / e(C, D) per ciphertext


def decrypt_batch_arrays(engine, folded, d_key, c_tilde, c, cy, cy_prime):
    """The same decryption on arrays (numpy, or CUDA tensors for HBM-resident ciphertexts): n ciphertexts under one
    (key, policy) plan, BASELINE config 4 at its stated size (2^16 ciphertexts x (2 x 256 + 1) pairs).

    c_tilde [n,384], c [n,64]; cy, cy_prime [n,l,64] with the l columns in the order of folded's leaf ids.  The factor
    1 / e(C, D) enters as e(C, -D) — one host-side negation of the key's D instead of n negations of C (both equal
    e(C, D)^-1 exactly) — so the G2 list [Dj^..., Dj'^..., -D] is the same for every ciphertext: one fixed-Q multi-pairing
    (2l+1 line evaluations and one final exponentiation per ciphertext) and one gt_mul."""
    leaf_ids, dj_hat, djp_hat = folded
    l = len(leaf_ids)
    q_list = np.concatenate([np.asarray(dj_hat).reshape(l, 128), np.asarray(djp_hat).reshape(l, 128), engine.g2_neg(d_key).reshape(1, 128)])
    if type(c).__module__.startswith("torch"):
        import torch
        n = c.numel() // 64
        P = torch.cat([cy.reshape(n, l, 64), cy_prime.reshape(n, l, 64), c.reshape(n, 1, 64)], dim=1).contiguous()
        Q = torch.from_numpy(np.ascontiguousarray(q_list)).to(c.device)
        return engine.gt_mul(c_tilde.reshape(n, 384).contiguous(), engine.multi_pair_fixed_q(P.reshape(-1), Q.reshape(-1)))
    n = np.asarray(c).size // 64
    P = np.concatenate([np.asarray(cy, dtype=np.uint8).reshape(n, l, 64), np.asarray(cy_prime, dtype=np.uint8).reshape(n, l, 64),
                        np.asarray(c, dtype=np.uint8).reshape(n, 1, 64)], axis=1)
    return engine.gt_mul(np.asarray(c_tilde, dtype=np.uint8).reshape(n, 384), engine.multi_pair_fixed_q(P, q_list))
