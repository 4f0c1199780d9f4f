"""Host-side mirror of the reference's hash package for the curve targets (hash/hash_to.go:113-119,169-175,204-210,
271-277; hash/hash_from_gt.go:5-8), batched: `bn254.HashToG1(msg, dst)` / `HashToG2(msg, dst)` of gnark-crypto as
published [EXT, parity unpinned] = RFC 9380 hash_to_curve with

    hash_to_field     expand_message_xmd(SHA-256), L = 48, two field elements                       }
    map_to_curve      Shallue-van de Woestijne, Z = 1, both elements, then one point addition      } on the GPU, one message
    clear_cofactor    G2 only: [x]P + psi([3x]P) + psi^2([x]P) + psi^3(P)                           } per lane (bn254.hash_to_g1/g2)

hash_to_g1 / hash_to_g2 below are the product path (csrc/xmd29.hip.hpp + h2c29.hip.hpp).  expand_message_xmd / hash_to_field
in this file are the hashlib restatement the tests hold against RFC 9380's K.1 vectors and against the device's field
elements; hash_to_g1_via_host_fields / _g2_ keep the older split (host hashing, device map) for that comparison.

The reference's ToField / BytesToField do not hash at all and are not reproduced here.
"""
import hashlib

import numpy as np

from . import bn254

P_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
L_BYTES = 48
DST_STRING_G1 = b"Hash String To Element In G1"
DST_BYTES_G1 = b"Hash Bytes To Element In G1"
DST_STRING_G2 = b"Hash String To Element In G2"
DST_BYTES_G2 = b"Hash Bytes To Element In G2"


def expand_message_xmd(msg, dst, n):
    """RFC 9380 §5.3.1 with SHA-256."""
    if len(dst) > 255:
        dst = hashlib.sha256(b"H2C-OVERSIZE-DST-" + dst).digest()
    ell = (n + 31) // 32
    if ell > 255 or n > 65535:
        raise ValueError("expand_message_xmd: requested length too large")
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(bytes(64) + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime).digest()
    bi = hashlib.sha256(b0 + b"\x01" + dst_prime).digest()
    out = [bi]
    for i in range(2, ell + 1):
        bi = hashlib.sha256(bytes(a ^ b for a, b in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
        out.append(bi)
    return b"".join(out)[:n]


def hash_to_field(msg, dst, count):
    """gnark fp.Hash(msg, dst, count): `count` base-field elements as Python ints."""
    u = expand_message_xmd(msg, dst, count * L_BYTES)
    return [int.from_bytes(u[L_BYTES * i:L_BYTES * (i + 1)], "big") % P_MOD for i in range(count)]


def _mont_rows(elements_per_msg, msgs, dst):
    """[n, elements_per_msg * 32] uint8: the field elements of every message in gnark's in-memory (Montgomery) layout."""
    rows = []
    for m in msgs:
        rows.append(b"".join((v * (1 << 256) % P_MOD).to_bytes(32, "little") for v in hash_to_field(m, dst, elements_per_msg)))
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(msgs), elements_per_msg * 32)


def hash_to_g1(msgs, dst):
    """bn254.HashToG1(msg, dst) for every message: [n, 64] G1Affine rows."""
    return bn254.hash_to_g1([bytes(m) for m in msgs], dst)


def hash_to_g2(msgs, dst):
    """bn254.HashToG2(msg, dst) for every message: [n, 128] G2Affine rows (E2 element j = base-field elements 2j, 2j+1)."""
    return bn254.hash_to_g2([bytes(m) for m in msgs], dst)


def hash_to_g1_via_host_fields(msgs, dst):
    msgs = [bytes(m) for m in msgs]
    if not msgs:
        return np.zeros((0, bn254.G1_BYTES), dtype=np.uint8)
    return bn254.map_to_g1(_mont_rows(2, msgs, dst))


def hash_to_g2_via_host_fields(msgs, dst):
    msgs = [bytes(m) for m in msgs]
    if not msgs:
        return np.zeros((0, bn254.G2_BYTES), dtype=np.uint8)
    return bn254.map_to_g2(_mont_rows(4, msgs, dst))


# the reference's four entry points (single value in, single point out)
def ToG1(s):
    return hash_to_g1([s.encode("utf-8")], DST_STRING_G1)[0]


def BytesToG1(b):
    return hash_to_g1([b], DST_BYTES_G1)[0]


def ToG2(s):
    return hash_to_g2([s.encode("utf-8")], DST_STRING_G2)[0]


def BytesToG2(b):
    return hash_to_g2([b], DST_BYTES_G2)[0]


def FromGT(gt):
    """hash.FromGT: GT.Bytes() (384 big-endian canonical bytes) of one value."""
    return bn254.gt_marshal(gt)[0].tobytes()
