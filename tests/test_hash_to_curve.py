"""Hash to curve on the CPU: RFC 9380's published expand_message_xmd vectors, the oracle against the
committed fixture, the host-side hashing of the product mirror (hash_to.py — plain hashlib, no GPU needed) against the
oracle, and the DEVICE code of csrc/h2c29.hip.hpp compiled for the host under the bounds harness against the fixture."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import bn254_py as o
from conftest import ROOT, load_golden

SO = os.path.join(ROOT, "tools", "libgpbc_bounds.so")


@pytest.fixture(scope="module")
def hc():
    src = os.path.join(ROOT, "tools", "bounds_check.cpp")
    hdrs = [os.path.join(ROOT, "gopairingbasedcryptography_amd", "csrc", f)
            for f in ("fe29.hip.hpp", "tower29.hip.hpp", "curve29.hip.hpp", "wire29.hip.hpp", "h2c29.hip.hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(f) > os.path.getmtime(SO) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-pthread", "-std=c++17", "-DGPBC_BOUNDS", "-shared", "-fPIC", "-o", SO, src])
    return ctypes.CDLL(SO)


def vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_rfc9380_expand_message_xmd_known_answers():
    """Published vectors (RFC 9380 Appendix K.1): pins the hashing half of the path, for the oracle and for the mirror."""
    from gopairingbasedcryptography_amd import hash_to
    g = load_golden("hash_to_curve.json")["rfc9380_expand_message_xmd_sha256"]
    for v in g["vectors"]:
        for fn in (o.expand_message_xmd, hash_to.expand_message_xmd):
            assert fn(v["msg"].encode(), g["dst"].encode(), v["len"]).hex() == v["uniform_bytes"], v["msg"]
    long_dst = b"x" * 300                                      # oversize DST is hashed first (RFC 9380 §5.3.3)
    assert o.expand_message_xmd(b"m", long_dst, 48) == hash_to.expand_message_xmd(b"m", long_dst, 48)
    with pytest.raises(ValueError):
        hash_to.expand_message_xmd(b"", b"d", 256 * 32)


def test_oracle_matches_fixture_and_group_facts():
    g = load_golden("hash_to_curve.json")
    assert str(o.SVDW_G1[0]) == g["svdw"]["Z_g1"] == "1" and [str(v) for v in o.SVDW_G2[0]] == g["svdw"]["Z_g2"]
    # c3 has sgn0 = 0 and c3^2 = -g(Z) * 3 Z^2 ; c1 = g(Z) = 4
    assert o.SVDW_G1[1] == 4 and o.SVDW_G1[3] % 2 == 0 and o.SVDW_G1[3] ** 2 % o.P == (-4 * 3) % o.P
    dsts = {k: v.encode() for k, v in g["dsts"].items()}
    for c in g["g1"]:
        u = o.hash_to_field_fp(c["msg"].encode(), dsts[c["dst"]], 2)
        assert [str(v) for v in u] == c["u"]
        pt = o.hash_to_g1(c["msg"].encode(), dsts[c["dst"]])
        assert o.g1_to_bytes(pt).hex() == c["point"] and o.g1_is_on_curve(pt)
    for c in g["g2"]:
        pt = o.hash_to_g2(c["msg"].encode(), dsts[c["dst"]])
        assert o.g2_to_bytes(pt).hex() == c["point"] and o.g2_in_subgroup(pt)
    # the map sends -u to the opposite point, so equal-and-opposite field elements give the point at infinity
    assert g["g1_fields"][4]["point"] == "00" * 64 and g["g2_fields"][3]["point"] == "00" * 128
    # different domain-separation tags give unrelated points
    assert g["g1"][0]["point"] != g["g1"][1]["point"]


def test_host_hashing_of_the_mirror_matches_oracle():
    from gopairingbasedcryptography_amd import hash_to
    for msg in (b"", b"abc", bytes(range(256)) * 3):
        for dst in (hash_to.DST_STRING_G1, hash_to.DST_BYTES_G2):
            assert hash_to.hash_to_field(msg, dst, 4) == o.hash_to_field_fp(msg, dst, 4)
    rows = hash_to._mont_rows(4, [b"abc"], hash_to.DST_STRING_G2)
    u = o.hash_to_field_fp2(b"abc", hash_to.DST_STRING_G2, 2)
    assert rows[0].tobytes() == o.f2_to_bytes(u[0]) + o.f2_to_bytes(u[1])


def _map(hc, g2, rows):
    w = 128 if g2 else 64
    u = np.frombuffer(b"".join(rows), dtype=np.uint8).copy()
    out = np.zeros((len(rows), w), dtype=np.uint8)
    hc.hc_map_fields(int(g2), vp(u), ctypes.c_size_t(len(rows)), vp(out))
    return out


def test_device_code_under_bounds_matches_fixture(hc):
    g = load_golden("hash_to_curve.json")
    cases = g["g1"][:6] + g["g1_fields"]
    out = _map(hc, False, [o.fp_to_mont_bytes(int(c["u"][0])) + o.fp_to_mont_bytes(int(c["u"][1])) for c in cases])
    for i, c in enumerate(cases):
        assert out[i].tobytes().hex() == c["point"], ("g1", i)
    cases = g["g2"][:3] + g["g2_fields"]
    f2 = lambda v: o.f2_to_bytes((int(v[0]), int(v[1])))
    out = _map(hc, True, [f2(c["u"][0]) + f2(c["u"][1]) for c in cases])
    for i, c in enumerate(cases):
        assert out[i].tobytes().hex() == c["point"], ("g2", i)
