#!/usr/bin/env python3
"""Generate tests/golden/wire.json (wire formats) from the big-integer oracle ALONE.

"Parity unpinned", as for the other fixtures: the encodings follow gnark-crypto v0.19.0 ecc/bn254 marshal.go as
published (flag bits, big-endian canonical coordinates, G2 as A1||A0, GT.Bytes() coefficient order); no
gnark-produced bytes exist in the reference to confirm them.  `mem` = gnark in-memory struct (Montgomery LE limbs),
`raw` = Marshal()/RawBytes(), `compressed` = Bytes().  Invalid cases carry the slot size and the expected `ok`.
Run:  python tests/golden/gen_wire_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import bn254_py as o  # noqa: E402


def off_subgroup_twist_point(seed):
    x = (seed, seed + 2)
    while True:
        y = o.f2_sqrt(o.f2_add(o.f2_mul(o.f2_sqr(x), x), o.B_G2))
        if y is not None and not o.g2_in_subgroup((x, y)):
            return (x, y)
        x = (x[0] + 1, x[1])


def main():
    ks = [1, 2, 3, o.R - 1, o.R - 2] + [o.bench_scalar("wire", i) for i in range(19)]
    g1 = [("[%d]g1" % k if k < 4 else "k=%s" % hex(k)[:12], o.g1_mul(o.G1_GEN, k)) for k in ks] + [("infinity", None)]
    g2 = [("[%d]g2" % k if k < 4 else "k=%s" % hex(k)[:12], o.g2_mul(o.G2_GEN, k)) for k in ks] + [("infinity", None)]
    doc = {"layout": "mem = gnark in-memory struct; raw/compressed = big-endian canonical wire form", "g1": [], "g2": [], "gt": []}
    for note, p in g1:
        doc["g1"].append({"note": note, "mem": o.g1_to_bytes(p).hex(), "raw": o.g1_marshal(p).hex(), "compressed": o.g1_marshal(p, True).hex()})
    for note, q in g2:
        doc["g2"].append({"note": note, "mem": o.g2_to_bytes(q).hex(), "raw": o.g2_marshal(q).hex(), "compressed": o.g2_marshal(q, True).hex()})
    gts = [("one", o.F12_ONE), ("e(g1,g2)", o.pair([o.G1_GEN], [o.G2_GEN]))]
    for i in range(4):
        gts.append(("e(P%d,Q%d)" % (i, i), o.pair([g1[5 + i][1]], [g2[5 + i][1]])))
    for note, g in gts:
        doc["gt"].append({"note": note, "mem": o.gt_to_bytes(g).hex(), "wire": o.gt_marshal(g).hex()})

    # ---- decoding cases that gnark rejects (ok = 0, zero output) or accepts in an unusual slot
    def case(kind, note, eb, wire):
        pt, ok = (o.g1_unmarshal if kind == "g1" else o.g2_unmarshal)(wire)
        mem = (o.g1_to_bytes if kind == "g1" else o.g2_to_bytes)(pt if ok else None)
        return {"note": note, "elem_bytes": eb, "wire": wire.hex(), "ok": int(ok), "mem": mem.hex()}
    be = lambda v: v.to_bytes(32, "big")
    flag = lambda b, f: bytes([b[0] | f]) + b[1:]
    x = 5
    while o.fp_sqrt((x ** 3 + 3) % o.P) is not None:
        x += 1
    p5 = g1[7][1]
    doc["decode_g1"] = [
        case("g1", "x = p (not canonical), uncompressed", 64, be(o.P) + be(2)),
        case("g1", "y = p + 2 would alias y = 2: not canonical", 64, be(1) + be(o.P + 2)),
        case("g1", "(1,3) not on the curve", 64, be(1) + be(3)),
        case("g1", "compressed x without a square root", 32, flag(be(x), 0x80)),
        case("g1", "compressed x >= p", 32, flag(be(o.P + 1 - (1 << 254) if o.P + 1 >= (1 << 254) else o.P + 1), 0xC0)),
        case("g1", "infinity flag with a stray bit", 32, bytes([0x40]) + bytes(30) + b"\x01"),
        case("g1", "infinity flag with payload bits in byte 0", 32, bytes([0x41]) + bytes(31)),
        case("g1", "compressed infinity", 32, bytes([0x40]) + bytes(31)),
        case("g1", "uncompressed flag in a 32-byte slot (short buffer)", 32, o.g1_marshal(p5)[:32]),
        case("g1", "compressed form in a 64-byte slot (gnark reads 32 bytes)", 64, o.g1_marshal(p5, True) + bytes(32)),
        case("g1", "compressed, larger-Y flag", 32, flag(be(p5[0]), 0xC0)),
        case("g1", "compressed, smaller-Y flag", 32, flag(be(p5[0]), 0x80)),
        case("g1", "uncompressed all zero = infinity", 64, bytes(64)),
    ]
    t = off_subgroup_twist_point(5)
    q5 = g2[7][1]
    xq = (7, 11)
    while o.f2_sqrt(o.f2_add(o.f2_mul(o.f2_sqr(xq), xq), o.B_G2)) is not None:
        xq = (xq[0] + 1, xq[1])
    doc["decode_g2"] = [
        case("g2", "twist point outside the order-r subgroup, uncompressed", 128, o.g2_marshal(t)),
        case("g2", "twist point outside the order-r subgroup, compressed", 64, o.g2_marshal(t, True)),
        case("g2", "not on the twist", 128, be(1) + be(2) + be(3) + be(4)),
        case("g2", "X.A0 = p (not canonical)", 128, be(q5[0][1]) + be(o.P) + be(q5[1][1]) + be(q5[1][0])),
        case("g2", "compressed x without a square root", 64, flag(be(xq[1]) + be(xq[0]), 0x80)),
        case("g2", "infinity flag with a stray bit in the second half", 64, bytes([0x40]) + bytes(62) + b"\x01"),
        case("g2", "compressed infinity", 64, bytes([0x40]) + bytes(63)),
        case("g2", "uncompressed flag in a 64-byte slot (short buffer)", 64, o.g2_marshal(q5)[:64]),
        case("g2", "compressed form in a 128-byte slot", 128, o.g2_marshal(q5, True) + bytes(64)),
        case("g2", "compressed, other sign flag", 64, bytes([o.g2_marshal(q5, True)[0] ^ 0x40]) + o.g2_marshal(q5, True)[1:]),
        case("g2", "uncompressed all zero = infinity", 128, bytes(128)),
    ]
    e = o.gt_marshal(gts[1][1])
    doc["decode_gt"] = [
        {"note": "coefficient C0.B0.A0 = p (not canonical)", "wire": (e[:352] + be(o.P)).hex(), "ok": 0, "mem": bytes(384).hex()},
        {"note": "coefficient C1.B2.A1 = 2^256 - 1", "wire": (b"\xff" * 32 + e[32:]).hex(), "ok": 0, "mem": bytes(384).hex()},
        {"note": "all zero (the zero of Fp12: SetBytes does not test group membership)", "wire": bytes(384).hex(), "ok": 1, "mem": bytes(384).hex()},
    ]
    path = os.path.join(HERE, "wire.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
