"""Synthetic workloads of SURVEY.md §8d at BASELINE.json's sizes, shared by bench.py and tests/ (measurement and test
infrastructure, not the product): deterministic inputs from the seeded SHA-256 scalar stream, built with the engine's own
kernels so that 2^16 .. 2^20 units take seconds, and valid instances of the reference's schemes so that every output can be
checked (decryption returns the message; a forged signature is rejected).

  config 3  aggregate(engine, n)        BLS aggregate verification of n signatures on one message point
  config 4  bsw07_instance(engine, ..)  BSW07 CP-ABE: one key, one n-of-n policy (256-of-256 or 16 x (16-of-16)), 2^16 ciphertexts
  config 5  afp25_instance(engine, ..)  AFP25 batched IBE: batches of B = 256 identities, 2^18 (ciphertext, identity) items

`engine` is the bn254 module (GPU).  Scalars the generator needs per ciphertext are arranged so that the host does O(1)
big-integer work per ciphertext, not O(policy size): all leaves but one take shares t_j * u_leaf (one variable-base scalar
multiplication [t_j] B_leaf with B_leaf = [u_leaf] g1 fixed), the last leaf takes the share that makes the Lagrange
combination come out at the ciphertext's secret s_j.  With n-of-n gates any leaf values are a valid sharing.
"""
import hashlib

import numpy as np

SEED = 0x424E323534
R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def bench_scalar(tag, i):
    pre = b"gpbc-bench/v1/" + tag.encode() + SEED.to_bytes(8, "little")
    return int.from_bytes(hashlib.sha256(pre + int(i).to_bytes(8, "little")).digest(), "big") % R_ORDER


def bench_scalar_ints(tag, start, n):
    pre = b"gpbc-bench/v1/" + tag.encode() + SEED.to_bytes(8, "little")
    sha, fb = hashlib.sha256, int.from_bytes
    return [fb(sha(pre + (start + j).to_bytes(8, "little")).digest(), "big") % R_ORDER for j in range(n)]


def ints_to_bytes(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8)


def bench_scalars(tag, start, n):
    """k(tag,i) = SHA-256("gpbc-bench/v1/" || tag || LE64(seed) || LE64(i)) mod r, as n x 32 LE bytes."""
    return ints_to_bytes(bench_scalar_ints(tag, start, n))


def _dev(engine, a, device):
    import torch
    return torch.from_numpy(np.array(a, dtype=np.uint8, copy=True)).to(device)


def _rows32(engine, a, device):
    """n x 32 scalar rows on the device (2-D, so that slicing takes rows, not bytes)."""
    return _dev(engine, a, device).reshape(-1, 32)


def _mul_shared(engine, base, scalars_bytes, device, g2=False):
    """[k_i] base for one shared base, device-resident result (the fixed-base window path from 16384 scalars on)."""
    fn = engine.g2_scalar_mul if g2 else engine.g1_scalar_mul
    return fn(_dev(engine, base, device), _dev(engine, scalars_bytes, device))


# ----------------------------------------------------------------------------------------------- config 3
def aggregate(engine, n, device, start=0):
    """n BLS key pairs x_i = k("x", i), pk_i = [x_i] g1, one message point H = [k("H", 0)] g2, sigma_i = [x_i] H; verifier
    scalars rho_i = k("rho", i) truncated to 128 bits.  Returns a dict of device tensors and the host-side scalar
    sum rho_i x_i mod r (so that B = [that] H can stand in for the G2 sum when only the G1 side is timed)."""
    g1, g2 = engine.generators()
    x = bench_scalar_ints("x", start, n)
    rho = [v & ((1 << 128) - 1) for v in bench_scalar_ints("rho", start, n)]
    H = engine.g2_scalar_mul(g2, [bench_scalar("H", 0)])[0]
    xb, rb = ints_to_bytes(x), ints_to_bytes(rho)
    pk = _mul_shared(engine, g1, xb, device)
    sigma = _mul_shared(engine, H, xb, device, g2=True)
    acc = 0
    for a, b in zip(rho, x):
        acc += a * b
    return {"pk": pk, "sigma": sigma, "rho": _rows32(engine, rb, device), "H": np.asarray(H), "g1": g1, "sum_rho_x": acc % R_ORDER, "n": n}


def aggregate_check(engine, A, B, H, g1):
    """e(A, H) * e(g1, -B) == 1  (signature/bls01_signature/bls_signature.go:78-84 on the aggregated points)."""
    P = np.concatenate([np.asarray(A, dtype=np.uint8).reshape(64), np.asarray(g1, dtype=np.uint8).reshape(64)])
    Q = np.concatenate([np.asarray(H, dtype=np.uint8).reshape(128), engine.g2_neg(np.asarray(B, dtype=np.uint8))])
    return bool(engine.pairing_check(P, Q))


# ----------------------------------------------------------------------------------------------- config 4
def bsw07_policy(kind):
    """"256of256": one 256-of-256 root over leaves with attributes 1..256; "16x16": a 16-of-16 root over sixteen 16-of-16
    gates.  Both use all 256 attributes and make a ciphertext 2 x 256 + 1 = 513 pairs."""
    from gopairingbasedcryptography_amd import bsw07
    L, T = bsw07.Leaf, bsw07.Threshold
    if kind == "256of256":
        tree = T(256, *[L(a) for a in range(1, 257)])
    elif kind == "16x16":
        tree = T(16, *[T(16, *[L(16 * g + i + 1) for i in range(16)]) for g in range(16)])
    else:
        raise ValueError(kind)
    bsw07.assign_leaf_ids(tree)
    return tree


def bsw07_instance(engine, kind, n_ct, device, start=0):
    """One BSW07 key holding all 256 attributes, one policy, n_ct ciphertexts (cpabe/bsw07/bsw07_cpabe.go:57-170) as
    device-resident arrays: c_tilde [n,384], c [n,64], cy / cy_prime [n,256,64] (columns in leaf-id order), the messages
    [n,384], and the key (D, Dj, Dj' per attribute).  H1(a) = [h_a] g1, H2(a) = [h_a] g2 with h_a from the scalar stream."""
    import torch
    from gopairingbasedcryptography_amd import bsw07
    g1, g2 = engine.generators()
    sc = lambda tag, i=0: bench_scalar("bsw07-" + tag, i)
    tree = bsw07_policy(kind)
    attrs = list(range(1, 257))
    alpha, beta, r = sc("alpha"), sc("beta"), sc("r")
    h = {a: sc("h", a) for a in attrs}
    rj = {a: sc("rj", a) for a in attrs}
    # key (bsw07_cpabe.go:97-130): D = [(alpha + r) / beta] g2, Dj = [r] g2 + [rj] H2(j) = [r + rj h_j] g2, Dj' = [rj] g2
    D = engine.g2_scalar_mul(g2, [(alpha + r) * pow(beta, -1, R_ORDER) % R_ORDER])[0]
    dj_rows = engine.g2_scalar_mul(g2, [(r + rj[a] * h[a]) % R_ORDER for a in attrs])
    djp_rows = engine.g2_scalar_mul(g2, [rj[a] for a in attrs])
    dj = {a: dj_rows[i] for i, a in enumerate(attrs)}
    dj_prime = {a: djp_rows[i] for i, a in enumerate(attrs)}
    plan = bsw07.decrypt_plan(tree, set(attrs))
    leaf_ids = sorted(plan)
    l = len(leaf_ids)
    # shares: leaf i < last gets t_j * u_i, the last leaf what the Lagrange combination needs to land on s_j
    u = {i: sc("u-" + kind, i) for i in leaf_ids[:-1]}
    U = sum(plan[i][1] * u[i] for i in leaf_ids[:-1]) % R_ORDER
    last = leaf_ids[-1]
    inv_last = pow(plan[last][1], -1, R_ORDER)
    s = bench_scalar_ints("bsw07-s", start, n_ct)
    t = bench_scalar_ints("bsw07-t", start, n_ct)
    m = bench_scalar_ints("bsw07-msg", start, n_ct)
    e_last = [(sj - tj * U) * inv_last % R_ORDER for sj, tj in zip(s, t)]
    tb, eb = _dev(engine, ints_to_bytes(t), device), _dev(engine, ints_to_bytes(e_last), device)
    cy = torch.empty((n_ct, l, 64), dtype=torch.uint8, device=device)
    cyp = torch.empty((n_ct, l, 64), dtype=torch.uint8, device=device)
    # bases B_i = [u_i] g1 and B'_i = [u_i h_att(i)] g1, then one shared-base batch of n_ct multiplications by t_j per leaf
    b_rows = engine.g1_scalar_mul(g1, [u[i] for i in leaf_ids[:-1]] + [1])
    bp_rows = engine.g1_scalar_mul(g1, [u[i] * h[plan[i][0]] % R_ORDER for i in leaf_ids[:-1]] + [h[plan[last][0]]])
    for col, i in enumerate(leaf_ids):
        k = eb if i == last else tb
        cy[:, col, :] = engine.g1_scalar_mul(_dev(engine, b_rows[col], device), k)
        cyp[:, col, :] = engine.g1_scalar_mul(_dev(engine, bp_rows[col], device), k)
    g1d = _dev(engine, g1, device)
    c = engine.g1_scalar_mul(g1d, _dev(engine, ints_to_bytes([sj * beta % R_ORDER for sj in s]), device))
    e = engine.pair_batch(g1, g2)                                                   # [1,384]
    ed = _dev(engine, e, device).reshape(-1)
    M = engine.gt_exp(ed.repeat(n_ct).reshape(n_ct, 384).contiguous(), _dev(engine, ints_to_bytes(m), device))
    c_tilde = engine.gt_exp(ed.repeat(n_ct).reshape(n_ct, 384).contiguous(),
                            _dev(engine, ints_to_bytes([(alpha * sj + mj) % R_ORDER for sj, mj in zip(s, m)]), device))   # e^(alpha s) * M
    return {"tree": tree, "plan": plan, "leaf_ids": leaf_ids, "D": np.asarray(D), "dj": dj, "dj_prime": dj_prime, "attrs": set(attrs),
            "c_tilde": c_tilde, "c": c, "cy": cy, "cy_prime": cyp, "msgs": M, "n": n_ct, "pairs_per_ct": 2 * l + 1}


def bsw07_ct_dict(inst, j):
    """Ciphertext j in the dict form of bsw07.decrypt_batch / the reference-shaped checker (host arrays)."""
    cy, cyp = inst["cy"][j].cpu().numpy(), inst["cy_prime"][j].cpu().numpy()
    return {"c_tilde": inst["c_tilde"][j].cpu().numpy(), "c": inst["c"][j].cpu().numpy(),
            "cy": {i: cy[col] for col, i in enumerate(inst["leaf_ids"])}, "cy_prime": {i: cyp[col] for col, i in enumerate(inst["leaf_ids"])}}


# ----------------------------------------------------------------------------------------------- config 5
def afp25_instance(engine, B, n_items, device, start=0):
    """AFP25 batched IBE (bibe/afp25_bibe/afp25_bibe.go:146-269, 327-334): n_items / B batches of B identities id = k("id", .),
    one (ciphertext, identity) item per identity, as device arrays: per item the batch digest D, the opening proof pi, the
    batch key sk [n,64 each], C1 [n,3,128], C2 [n,384] and the messages [n,384].  The generator knows tau, so D = [f(tau)] g1
    and pi_i = [f(tau) / (tau - id_i)] g1 come from scalars (one batched modular inversion); afp25.digest / quotient_by_root /
    commit_g1_many compute the same values from the SRS, which the tests check on whole batches."""
    g1, g2 = engine.generators()
    sc = lambda tag, i=0: bench_scalar("afp25-" + tag, i)
    assert n_items % B == 0
    nb = n_items // B
    msk, tau, ht = sc("msk"), sc("tau"), sc("ht")
    ids = bench_scalar_ints("afp25-id", start, n_items)
    r1 = bench_scalar_ints("afp25-r1", start, n_items)
    r2 = bench_scalar_ints("afp25-r2", start, n_items)
    m = bench_scalar_ints("afp25-msg", start, n_items)
    diff = [(tau - i) % R_ORDER for i in ids]
    # batched inversion of (tau - id)
    pref, acc = [], 1
    for d in diff:
        pref.append(acc)
        acc = acc * d % R_ORDER
    inv_acc = pow(acc, -1, R_ORDER)
    inv = [0] * n_items
    for j in range(n_items - 1, -1, -1):
        inv[j] = inv_acc * pref[j] % R_ORDER
        inv_acc = inv_acc * diff[j] % R_ORDER
    f_tau = []
    for b in range(nb):
        a = 1
        for d in diff[b * B:(b + 1) * B]:
            a = a * d % R_ORDER
        f_tau.append(a)
    ft_item = [f_tau[j // B] for j in range(n_items)]
    g1d, g2d = _dev(engine, g1, device), _dev(engine, g2, device)
    mulg1 = lambda vals: engine.g1_scalar_mul(g1d, _dev(engine, ints_to_bytes(vals), device))
    mulg2 = lambda vals: engine.g2_scalar_mul(g2d, _dev(engine, ints_to_bytes(vals), device))
    D = mulg1(ft_item)
    pi = mulg1([f * iv % R_ORDER for f, iv in zip(ft_item, inv)])
    sk = mulg1([msk * (f + ht) % R_ORDER for f in ft_item])                           # [msk](D + h(t)), h(t) = [ht] g1
    import torch
    C1 = torch.stack([mulg2([(a + msk * b) % R_ORDER for a, b in zip(r1, r2)]),        # c1 = r^T A, A = [[g2, [id - tau] g2, 0], [[msk] g2, 0, -g2]]
                      mulg2([a * ((i - tau) % R_ORDER) % R_ORDER for a, i in zip(r1, ids)]),
                      mulg2([(-b) % R_ORDER for b in r2])], dim=1).contiguous()
    e = _dev(engine, engine.pair_batch(g1, g2), device).reshape(-1)
    E = e.repeat(n_items).reshape(n_items, 384).contiguous()
    M = engine.gt_exp(E, _dev(engine, ints_to_bytes(m), device))
    C2 = engine.gt_exp(E, _dev(engine, ints_to_bytes([(mj - ht * msk % R_ORDER * b) % R_ORDER for mj, b in zip(m, r2)]), device))   # e(h(t), [msk] g2)^-r2 * M
    return {"D": D, "pi": pi, "sk": sk, "C1": C1, "C2": C2, "msgs": M, "ids": ids, "tau": tau, "B": B, "n": n_items,
            "msk": msk, "ht": ht, "g1": g1, "g2": g2}


def afp25_srs(engine, inst, n_powers=None):
    """[tau^j] g1, j = 1..B: the commitment bases the host planner (afp25.digest / commit_g1_many) works from."""
    B = n_powers or inst["B"]
    return engine.g1_scalar_mul(inst["g1"], [pow(inst["tau"], j, R_ORDER) for j in range(1, B + 1)])
