#!/usr/bin/env python3
"""Generate tests/golden/*.json from the big-integer oracle (oracle/bn254_py.py) ALONE.

The reference cannot run here (Go module, toolchain absent; its arithmetic dependency gnark-crypto
v0.19.0 is not vendored — SURVEY.md §8c) and its tests hold no known-answer vectors, so these
fixtures are "parity unpinned": they pin the C restatement and the HIP kernels to the textbook
definition in bn254_py.py, not to bytes produced by gnark-crypto.  e(g1,g2) is also written in
canonical big-endian form (gnark GT.Bytes() order) so ONE future gnark-produced vector can confirm
or refute the layout/exponent conventions in a single comparison.

All byte strings are hex of gnark in-memory layouts (fp.Element = 4 LE u64 limbs, Montgomery form).
Run:  python tests/golden/gen_golden.py      (about a minute)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import bn254_py as o  # noqa: E402

N_PAIR = 48
N_MUL = 48


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote", path)


def synth_p(i): return o.g1_mul(o.G1_GEN, o.bench_scalar("P", i))
def synth_q(i): return o.g2_mul(o.G2_GEN, o.bench_scalar("Q", i))


def main():
    o.self_check()
    # ---------------------------------------------------------------- pairings
    cases = []
    pts = [(o.G1_GEN, o.G2_GEN, "generators"),
           (o.g1_neg(o.G1_GEN), o.G2_GEN, "-g1,g2"),
           (None, o.G2_GEN, "P=inf"),
           (o.G1_GEN, None, "Q=inf"),
           (None, None, "both inf")]
    for i in range(N_PAIR):
        pts.append((synth_p(i), synth_q(i), "bench stream i=%d" % i))
    for p, q, note in pts:
        gt = o.pair([p], [q])
        cases.append({"note": note, "P": o.g1_to_bytes(p).hex(), "Q": o.g2_to_bytes(q).hex(),
                      "GT": o.gt_to_bytes(gt).hex()})
    e = o.pair([o.G1_GEN], [o.G2_GEN])
    flat = [x for half in e for c in half for x in c]
    dump("pairing.json", {
        "layout": "gnark in-memory (Montgomery LE limbs); P 64B, Q 128B, GT 384B",
        "e_g1_g2_canonical_be_GTBytes_order": o.gt_to_canonical_bytes(e).hex(),
        "e_g1_g2_decimal_C0B0A0_to_C1B2A1": [str(x) for x in flat],
        "cases": cases})
    # ---------------------------------------------------------------- multi-pairings (segments)
    segs = []
    k = 0
    shapes = [1, 2, 3, 4, 2, 5, 8, 1, 3, 2, 16, 2]
    for s, m in enumerate(shapes):
        ps, qs = [], []
        for j in range(m):
            ps.append(synth_p(100 + k)); qs.append(synth_q(100 + k)); k += 1
        note = "%d pairs" % m
        if s == 4:      # e(P,Q) * e(-P,Q) = 1 (BLS-verify shape, PairingCheck true)
            ps[1] = o.g1_neg(ps[0]); qs[1] = qs[0]; note += ", cancelling"
        if s == 5:      # infinity entries are skipped
            ps[2] = None; qs[4] = None; note += ", with infinities"
        if s == 9:      # all-infinity segment -> one
            ps = [None, None]; note += ", all infinity"
        if s == 11:     # BLS: e([x]g1, H) * e(g1, -[x]H) = 1
            x = o.bench_scalar("x", 0)
            H = synth_q(999)
            ps = [o.g1_mul(o.G1_GEN, x), o.G1_GEN]
            qs = [H, o.g2_neg(o.g2_mul(H, x))]
            note += ", BLS verify identity"
        gt = o.pair(ps, qs)
        segs.append({"note": note, "P": [o.g1_to_bytes(p).hex() for p in ps],
                     "Q": [o.g2_to_bytes(q).hex() for q in qs], "GT": o.gt_to_bytes(gt).hex(),
                     "is_one": gt == o.F12_ONE})
    dump("multi_pair.json", {"segments": segs})
    # ---------------------------------------------------------------- scalar multiplications
    edge_scalars = [0, 1, 2, o.R - 1, o.R, o.R + 5, (1 << 256) - 1, 1 << 255, 3]
    g1c, g2c = [], []
    for i in range(N_MUL):
        if i < len(edge_scalars):
            s, note = edge_scalars[i], "edge scalar"
        else:
            s, note = o.bench_scalar("s", i), "bench stream"
        if i == N_MUL - 1:
            b1, b2, note = None, None, "base = infinity"
        elif i == N_MUL - 2:
            b1, b2, note = o.G1_GEN, o.G2_GEN, "generator base"
        else:
            b1, b2 = synth_p(i), synth_q(i)
        g1c.append({"note": note, "base": o.g1_to_bytes(b1).hex(), "scalar": o.scalar_to_bytes(s).hex(),
                    "out": o.g1_to_bytes(o.g1_mul(b1, s) if b1 else None).hex()})
        g2c.append({"note": note, "base": o.g2_to_bytes(b2).hex(), "scalar": o.scalar_to_bytes(s).hex(),
                    "out": o.g2_to_bytes(o.g2_mul(b2, s) if b2 else None).hex()})
    two_g1 = o.g1_mul(o.G1_GEN, 2)
    dump("g1_scalar_mul.json", {"two_g1_decimal": [str(two_g1[0]), str(two_g1[1])], "cases": g1c})
    dump("g2_scalar_mul.json", {"cases": g2c})
    # ---------------------------------------------------------------- GT ops
    gts = [o.pair([synth_p(i)], [synth_q(i)]) for i in range(4)]
    gexp = []
    for i, kx in enumerate([0, 1, 2, o.R - 1, o.R, o.bench_scalar("k", 0), o.bench_scalar("k", 1),
                            (1 << 256) - 1]):
        x = gts[i % 4]
        # 256-bit plain exponent; x has order r so the oracle may reduce, the kernels need not
        gexp.append({"x": o.gt_to_bytes(x).hex(), "k": o.scalar_to_bytes(kx).hex(),
                     "out": o.gt_to_bytes(o.gt_exp(x, kx % o.R)).hex()})
    gbin = []
    for i in range(4):
        a, b = gts[i], gts[(i + 1) % 4]
        gbin.append({"a": o.gt_to_bytes(a).hex(), "b": o.gt_to_bytes(b).hex(),
                     "mul": o.gt_to_bytes(o.f12_mul(a, b)).hex(),
                     "div": o.gt_to_bytes(o.f12_mul(a, o.f12_inv(b))).hex(),
                     "inv_a": o.gt_to_bytes(o.f12_inv(a)).hex()})
    dump("gt_ops.json", {"exp": gexp, "binary": gbin})


if __name__ == "__main__":
    main()
