"""Consumer of the pin hand-off (tools/gnark_vectors/main.go -> tests/golden/gnark_vectors.json).

PARITY UNPINNED until that file exists: the reference's arithmetic lives in gnark-crypto v0.19.0 (go.mod:5), which cannot run
in the build image (no Go toolchain) and whose source is not in /root/reference; none of the reference's 147 tests holds a
G1 / G2 / GT byte.  A maintainer with Go runs the harness once and drops the JSON into tests/golden/; these tests then check
the big-integer oracle (CPU) and the engine (GPU, through the C ABI) against every value gnark printed: in-memory limbs,
Marshal() / Bytes() encodings, Pair (incl. a 2-pair product and PairingCheck), scalar multiplications, GT.Exp / Mul / Div /
Inverse and HashToG1 / HashToG2 under the reference's DSTs.  Without the file they skip with "parity unpinned".

The checker itself is exercised either way: test_consumer_on_an_oracle_made_file builds a file of the same shape from the
oracle (what main.go would print if gnark agrees with the oracle) and runs the same comparison over it.
"""
import json
import os

import numpy as np
import pytest

import bn254_py as o
from conftest import GOLDEN

VECTORS = os.path.join(GOLDEN, "gnark_vectors.json")
KS = ["1", "2", "3", "65537", "1311768467463790320",
      "6296462850587514219860166612309923493513421339716397012889107043265683424215",
      str(o.R - 1)]
H2C_MSGS = ["abc", "", "GoPairingBasedCryptography"]
DST_G1 = ["Hash String To Element In G1", "Hash Bytes To Element In G1"]                      # hash/hash_to.go:114,170
DST_G2 = ["Hash String To Element In G2", "Hash Bytes To Element In G2", "signature SigmaSignature"]   # :205,272; bls_signature_demo.go:25


def _g1(pt):
    return {"raw": o.g1_to_bytes(pt).hex(), "marshal": o.g1_marshal(pt).hex(), "bytes": o.g1_marshal(pt, True).hex()}


def _g2(pt):
    return {"raw": o.g2_to_bytes(pt).hex(), "marshal": o.g2_marshal(pt).hex(), "bytes": o.g2_marshal(pt, True).hex()}


def _gt(x):
    return {"raw": o.gt_to_bytes(x).hex(), "bytes": o.gt_marshal(x).hex()}


def oracle_made_vectors():
    """The file main.go prints, computed by the oracle instead of gnark."""
    p1 = [o.g1_mul(o.G1_GEN, int(k)) for k in KS]
    p2 = [o.g2_mul(o.G2_GEN, int(k)) for k in KS]
    e = o.pair([o.G1_GEN], [o.G2_GEN])
    e35 = o.pair([p1[2]], [p2[3]])
    v = {"gnark_crypto_version": "oracle (bn254_py), not gnark", "g1": _g1(o.G1_GEN), "g2": _g2(o.G2_GEN), "pair_g1_g2": _gt(e),
         "scalar_mul": [{"k": k, "g1": _g1(a), "g2": _g2(b)} for k, a, b in zip(KS, p1, p2)],
         "pair_3_65537": _gt(e35), "pair_product_idx_1_5__4_2": _gt(o.pair([p1[1], p1[4]], [p2[5], p2[2]])),
         "pairing_check_true": True,
         "gt_exp_pair_by_k5": _gt(o.gt_exp(e, int(KS[5]))), "gt_mul": _gt(o.f12_mul(e, e35)),
         "gt_div": _gt(o.f12_mul(e, o.f12_inv(e35))), "gt_inverse": _gt(o.f12_inv(e35)), "hash_to_curve": [],
         "fr_one_raw": o.fr_to_mont_bytes(1).hex()}
    for msg in H2C_MSGS:
        for dst in DST_G1:
            v["hash_to_curve"].append({"group": "g1", "msg": msg, "dst": dst, "point": _g1(o.hash_to_g1(msg.encode(), dst.encode()))})
        for dst in DST_G2:
            v["hash_to_curve"].append({"group": "g2", "msg": msg, "dst": dst, "point": _g2(o.hash_to_g2(msg.encode(), dst.encode()))})
    return v


def check_oracle(v):
    """Every gnark value against the big-integer oracle."""
    assert v["g1"] == _g1(o.G1_GEN) and v["g2"] == _g2(o.G2_GEN), "generators / Montgomery layout / Marshal flags"
    e = o.pair([o.G1_GEN], [o.G2_GEN])
    assert v["pair_g1_g2"]["raw"] == o.gt_to_bytes(e).hex(), "Pair(g1,g2): final-exponent cofactor or tower basis order"
    assert v["pair_g1_g2"]["bytes"] == o.gt_marshal(e).hex(), "GT.Bytes() coefficient order"
    p1, p2 = [], []
    for row in v["scalar_mul"]:
        a, b = o.g1_mul(o.G1_GEN, int(row["k"])), o.g2_mul(o.G2_GEN, int(row["k"]))
        assert row["g1"] == _g1(a) and row["g2"] == _g2(b), "scalar multiplication by " + row["k"]
        p1.append(a), p2.append(b)
    e35 = o.pair([p1[2]], [p2[3]])
    assert v["pair_3_65537"] == _gt(e35)
    assert v["pair_product_idx_1_5__4_2"] == _gt(o.pair([p1[1], p1[4]], [p2[5], p2[2]]))
    assert v["pairing_check_true"] is True
    assert v["gt_exp_pair_by_k5"] == _gt(o.gt_exp(e, int(v["scalar_mul"][5]["k"])))
    assert v["gt_mul"] == _gt(o.f12_mul(e, e35)) and v["gt_div"] == _gt(o.f12_mul(e, o.f12_inv(e35)))
    assert v["gt_inverse"] == _gt(o.f12_inv(e35))
    for h in v["hash_to_curve"]:
        if h["group"] == "g1":
            assert h["point"] == _g1(o.hash_to_g1(h["msg"].encode(), h["dst"].encode())), ("HashToG1", h["msg"], h["dst"])
        else:
            assert h["point"] == _g2(o.hash_to_g2(h["msg"].encode(), h["dst"].encode())), ("HashToG2 (Z, cofactor clearing)", h["msg"], h["dst"])
    assert v["fr_one_raw"] == o.fr_to_mont_bytes(1).hex()


def check_engine(v, eng):
    """Every gnark value against the HIP path, through the C ABI."""
    from gopairingbasedcryptography_amd import hash_to
    hx = lambda s: np.frombuffer(bytes.fromhex(s), dtype=np.uint8)
    g1, g2 = eng.generators()
    assert g1.tobytes().hex() == v["g1"]["raw"] and g2.tobytes().hex() == v["g2"]["raw"]
    assert eng.pair_batch(g1, g2)[0].tobytes().hex() == v["pair_g1_g2"]["raw"]
    ks = [int(r["k"]) for r in v["scalar_mul"]]
    P, Q = eng.g1_scalar_mul(g1, ks), eng.g2_scalar_mul(g2, ks)
    for i, row in enumerate(v["scalar_mul"]):
        assert P[i].tobytes().hex() == row["g1"]["raw"] and Q[i].tobytes().hex() == row["g2"]["raw"], row["k"]
    assert eng.g1_marshal(P).tobytes().hex() == "".join(r["g1"]["marshal"] for r in v["scalar_mul"])
    assert eng.g1_marshal(P, compressed=True).tobytes().hex() == "".join(r["g1"]["bytes"] for r in v["scalar_mul"])
    assert eng.g2_marshal(Q).tobytes().hex() == "".join(r["g2"]["marshal"] for r in v["scalar_mul"])
    assert eng.g2_marshal(Q, compressed=True).tobytes().hex() == "".join(r["g2"]["bytes"] for r in v["scalar_mul"])
    back, ok = eng.g2_unmarshal(hx("".join(r["g2"]["bytes"] for r in v["scalar_mul"])), elem_bytes=64)
    assert ok.all() and (back == Q).all()
    e = hx(v["pair_g1_g2"]["raw"])
    e35 = eng.pair_batch(P[2], Q[3])[0]
    assert e35.tobytes().hex() == v["pair_3_65537"]["raw"]
    assert eng.pair(np.stack([P[1], P[4]]), np.stack([Q[5], Q[2]])).tobytes().hex() == v["pair_product_idx_1_5__4_2"]["raw"]
    assert eng.gt_marshal(e)[0].tobytes().hex() == v["pair_g1_g2"]["bytes"]
    assert eng.gt_exp(e, [ks[5]])[0].tobytes().hex() == v["gt_exp_pair_by_k5"]["raw"]
    assert eng.gt_mul(e, e35)[0].tobytes().hex() == v["gt_mul"]["raw"]
    assert eng.gt_div(e, e35)[0].tobytes().hex() == v["gt_div"]["raw"]
    assert eng.gt_inverse(e35)[0].tobytes().hex() == v["gt_inverse"]["raw"]
    for h in v["hash_to_curve"]:
        fn = hash_to.hash_to_g1 if h["group"] == "g1" else hash_to.hash_to_g2
        assert fn([h["msg"].encode()], h["dst"].encode())[0].tobytes().hex() == h["point"]["raw"], (h["group"], h["msg"], h["dst"])


def load_vectors():
    if not os.path.exists(VECTORS):
        pytest.skip("parity unpinned: tests/golden/gnark_vectors.json absent — run tools/gnark_vectors/main.go where Go and "
                    "gnark-crypto v0.19.0 exist and commit its output")
    with open(VECTORS) as f:
        return json.load(f)


def test_harness_source_is_committed():
    src = open(os.path.join(os.path.dirname(GOLDEN), "..", "tools", "gnark_vectors", "main.go")).read()
    for needle in ("bn254.Pair(", "ScalarMultiplication(", "HashToG1(", "HashToG2(", ".Marshal()", ".Bytes()", "PairingCheck("):
        assert needle in src, needle
    for dst in DST_G1 + DST_G2:
        assert dst in src, dst
    for k in KS:
        assert '"%s"' % k in src, k


def test_consumer_on_an_oracle_made_file():
    v = json.loads(json.dumps(oracle_made_vectors()))
    check_oracle(v)
    bad = json.loads(json.dumps(v))
    raw = bytearray(bytes.fromhex(bad["pair_g1_g2"]["raw"]))
    raw[5] ^= 1
    bad["pair_g1_g2"]["raw"] = raw.hex()
    with pytest.raises(AssertionError):
        check_oracle(bad)


def test_oracle_against_gnark_vectors():
    check_oracle(load_vectors())


@pytest.mark.gpu
def test_engine_against_gnark_vectors():
    v = load_vectors()
    from gopairingbasedcryptography_amd import _build, bn254
    _build.build_library()
    bn254.init(0)
    check_engine(v, bn254)


@pytest.mark.gpu
def test_engine_consumer_on_an_oracle_made_file():
    from gopairingbasedcryptography_amd import _build, bn254
    _build.build_library()
    bn254.init(0)
    check_engine(json.loads(json.dumps(oracle_made_vectors())), bn254)
