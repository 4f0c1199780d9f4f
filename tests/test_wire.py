"""Wire formats on the CPU: the big-integer oracle against the committed fixture, structural facts of
the published gnark encoding, and the DEVICE code of csrc/wire29.hip.hpp compiled for the host under the bounds harness
(tools/bounds_check.cpp) against the same fixture.  The GPU parity tests are in test_gpu_parity.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import bn254_py as o
from conftest import ROOT, cat, hx, load_golden

SO = os.path.join(ROOT, "tools", "libgpbc_bounds.so")


@pytest.fixture(scope="module")
def hc():
    src = os.path.join(ROOT, "tools", "bounds_check.cpp")
    hdrs = [os.path.join(ROOT, "gopairingbasedcryptography_amd", "csrc", f)
            for f in ("fe29.hip.hpp", "tower29.hip.hpp", "tower29_pair.hip.hpp", "curve29.hip.hpp", "pairing29.hip.hpp", "pairing29_pair.hip.hpp", "wire29.hip.hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(f) > os.path.getmtime(SO) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-pthread", "-std=c++17", "-DGPBC_BOUNDS", "-shared", "-fPIC", "-o", SO, src])
    return ctypes.CDLL(SO)


def vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_oracle_matches_fixture():
    g = load_golden("wire.json")
    for c in g["g1"]:
        p = o.g1_from_bytes(bytes.fromhex(c["mem"]))
        assert o.g1_marshal(p).hex() == c["raw"] and o.g1_marshal(p, True).hex() == c["compressed"], c["note"]
        for enc in (c["raw"], c["compressed"]):
            assert o.g1_unmarshal(bytes.fromhex(enc)) == (p, True), c["note"]
    for c in g["g2"]:
        q = o.g2_from_bytes(bytes.fromhex(c["mem"]))
        assert o.g2_marshal(q).hex() == c["raw"] and o.g2_marshal(q, True).hex() == c["compressed"], c["note"]
        for enc in (c["raw"], c["compressed"]):
            assert o.g2_unmarshal(bytes.fromhex(enc)) == (q, True), c["note"]
    for c in g["gt"]:
        v = o.gt_from_bytes(bytes.fromhex(c["mem"]))
        assert o.gt_marshal(v).hex() == c["wire"]
        assert o.gt_unmarshal(bytes.fromhex(c["wire"])) == (v, True)
    for kind, fn, to_mem in (("decode_g1", o.g1_unmarshal, o.g1_to_bytes), ("decode_g2", o.g2_unmarshal, o.g2_to_bytes)):
        for c in g[kind]:
            pt, ok = fn(bytes.fromhex(c["wire"]))
            assert int(ok) == c["ok"] and to_mem(pt if ok else None).hex() == c["mem"], c["note"]
    for c in g["decode_gt"]:
        v, ok = o.gt_unmarshal(bytes.fromhex(c["wire"]))
        assert int(ok) == c["ok"], c["note"]


def test_published_structure():
    """Facts of gnark's encoding that do not depend on this repository's arithmetic."""
    # g1 = (1, 2): y = 2 is the smaller of {2, p - 2}
    assert o.g1_marshal(o.G1_GEN, True) == bytes([0x80]) + bytes(30) + b"\x01"
    assert o.g1_marshal(o.g1_neg(o.G1_GEN), True) == bytes([0xC0]) + bytes(30) + b"\x01"
    assert o.g1_marshal(o.G1_GEN) == bytes(31) + b"\x01" + bytes(31) + b"\x02"
    assert o.g1_marshal(None) == bytes(64) and o.g1_marshal(None, True) == bytes([0x40]) + bytes(31)
    assert o.g2_marshal(None, True) == bytes([0x40]) + bytes(63)
    # G2 writes the imaginary part first
    raw = o.g2_marshal(o.G2_GEN)
    assert int.from_bytes(raw[0:32], "big") == o.G2_GEN[0][1] and int.from_bytes(raw[32:64], "big") == o.G2_GEN[0][0]
    # GT: one = ... 00 01 in the LAST 32 bytes (C0.B0.A0 is written last)
    assert o.gt_marshal(o.F12_ONE) == bytes(383) + b"\x01"
    # the two compressed forms of one x decode to opposite points
    x = o.g1_mul(o.G1_GEN, 77)
    a, _ = o.g1_unmarshal(bytes([0x80 | x[0].to_bytes(32, "big")[0]]) + x[0].to_bytes(32, "big")[1:])
    b, _ = o.g1_unmarshal(bytes([0xC0 | x[0].to_bytes(32, "big")[0]]) + x[0].to_bytes(32, "big")[1:])
    assert a == o.g1_neg(b) and x in (a, b)


def _encode(hc, kind, mem, n, compressed, width):
    out = np.zeros((n, width), dtype=np.uint8)
    mem = mem.copy()
    hc.hc_wire_encode(kind, vp(mem), ctypes.c_size_t(n), int(compressed), vp(out))
    return out


def _decode(hc, kind, wire, elem_bytes, n, width):
    out = np.full((n, width), 0xAA, dtype=np.uint8)
    ok = np.full(n, 7, dtype=np.uint8)
    wire = wire.copy()
    hc.hc_wire_decode(kind, vp(wire), int(elem_bytes), ctypes.c_size_t(n), vp(out), vp(ok))
    return out, ok


def test_device_code_under_bounds_matches_fixture(hc):
    g = load_golden("wire.json")
    for kind, key, mem_w, raw_w, comp_w in ((0, "g1", 64, 64, 32), (1, "g2", 128, 128, 64)):
        cs = g[key]
        mem = cat([c["mem"] for c in cs])
        for comp, field, w in ((False, "raw", raw_w), (True, "compressed", comp_w)):
            enc = _encode(hc, kind, mem, len(cs), comp, w)
            for i, c in enumerate(cs):
                assert enc[i].tobytes().hex() == c[field], (key, field, c["note"])
            dec, ok = _decode(hc, kind, cat([c[field] for c in cs]), w, len(cs), mem_w)
            assert ok.tolist() == [1] * len(cs)
            assert dec.tobytes() == mem.tobytes(), (key, field)
        for c in g["decode_" + key]:
            dec, ok = _decode(hc, kind, hx(c["wire"]), c["elem_bytes"], 1, mem_w)
            assert int(ok[0]) == c["ok"] and dec[0].tobytes().hex() == c["mem"], (key, c["note"])
    cs = g["gt"]
    mem = cat([c["mem"] for c in cs])
    enc = _encode(hc, 2, mem, len(cs), False, 384)
    assert enc.tobytes().hex() == "".join(c["wire"] for c in cs)
    dec, ok = _decode(hc, 2, cat([c["wire"] for c in cs]), 384, len(cs), 384)
    assert ok.all() and dec.tobytes() == mem.tobytes()
    for c in g["decode_gt"]:
        dec, ok = _decode(hc, 2, hx(c["wire"]), 384, 1, 384)
        assert int(ok[0]) == c["ok"] and dec[0].tobytes().hex() == c["mem"], c["note"]


def test_random_corruption_agrees_with_oracle(hc):
    """Flip bits of valid encodings: the device code and the oracle must accept / reject the same buffers and agree on
    every accepted point."""
    rng = np.random.default_rng(254)
    g = load_golden("wire.json")
    for kind, key, mem_w, unm, to_mem in ((0, "g1", 64, o.g1_unmarshal, o.g1_to_bytes), (1, "g2", 128, o.g2_unmarshal, o.g2_to_bytes)):
        for field in ("raw", "compressed"):
            bufs = []
            for c in g[key][:12]:
                b = bytearray(bytes.fromhex(c[field]))
                for _ in range(int(rng.integers(1, 3))):
                    b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
                bufs.append(bytes(b))
            w = len(bufs[0])
            dec, ok = _decode(hc, kind, np.frombuffer(b"".join(bufs), dtype=np.uint8), w, len(bufs), mem_w)
            for i, b in enumerate(bufs):
                pt, good = unm(b)
                assert int(good) == int(ok[i]), (key, field, i)
                assert dec[i].tobytes() == to_mem(pt if good else None), (key, field, i)


def test_g2_subgroup_test_agrees_with_the_definition(hc):
    """The device code decides subgroup membership by gnark's endomorphism identity
    [x+1]Q + psi([x]Q) + psi^2([x]Q) = psi^3([2x]Q); the oracle by the definition [r]Q = infinity.  They must agree on
    subgroup points, random twist points, points of the cofactor group (order dividing h2 = 2p - r, including its small
    prime factor 10069) and sums of a subgroup point and a cofactor point."""
    import random
    random.seed(2024)
    h2 = 2 * o.P - o.R
    assert h2 % 10069 == 0

    def rand_twist():
        while True:
            x = (random.randrange(o.P), random.randrange(o.P))
            y = o.f2_sqrt(o.f2_add(o.f2_mul(o.f2_sqr(x), x), o.B_G2))
            if y is not None:
                return (x, y)

    pts = [o.g2_mul(o.G2_GEN, random.randrange(1, o.R)) for _ in range(6)]
    pts += [rand_twist() for _ in range(6)]
    cof = [o.g2_mul_plain(rand_twist(), o.R) for _ in range(4)]                      # order divides h2
    small = [t for t in (o.g2_mul_plain(rand_twist(), (h2 // 10069) * o.R) for _ in range(3)) if t is not None]
    pts += cof + small
    pts += [o.g2_add(pts[i], cof[i]) for i in range(4)]                              # subgroup + cofactor component
    want = [int(o.g2_in_subgroup(p)) for p in pts]
    assert want[:6] == [1] * 6 and not any(want[6:])
    for comp, w in ((False, 128), (True, 64)):
        wire = np.frombuffer(b"".join(o.g2_marshal(p, comp) for p in pts), dtype=np.uint8)
        dec, ok = _decode(hc, 1, wire, w, len(pts), 128)
        assert ok.tolist() == want, comp
        for i, p in enumerate(pts):
            assert dec[i].tobytes() == o.g2_to_bytes(p if want[i] else None)
