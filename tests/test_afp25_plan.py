"""BASELINE config 5 semantics (bibe/afp25_bibe decrypt) through the host planner gopairingbasedcryptography_amd/afp25.py:
quotient polynomials by synthetic division, pi by scalar-mult + point-sum, one 3-pair multi-pairing per item.  The result
must be the encrypted message AND bit-identical to the reference-shaped evaluation (afp25_bibe.go:369-418)."""
import numpy as np
import pytest

import bn254_py as o
from afp25_fixture import Instance
from gopairingbasedcryptography_amd import afp25
from test_bsw07_plan import OracleEngine


class Eng(OracleEngine):
    def g1_sum(self, p): return self.o.g1_sum(p)
    def gt_div(self, a, b): return self.o.gt_div(a, b)


def test_polynomials():
    # (x-1)(x-2) = 2 - 3x + x^2  (bibe/gwww25_bibe/gwww25_bibe_test.go:400-428)
    assert afp25.poly_from_roots([1, 2]) == [2, o.R - 3, 1]
    f = afp25.poly_from_roots([5, 7, 11, 13])
    assert afp25.quotient_by_root(f, 7) == afp25.poly_from_roots([5, 11, 13])
    with pytest.raises(ValueError, match="identity not found"):
        afp25.quotient_by_root(f, 8)


def test_batched_decrypt_matches_reference_shape(oracle):
    eng = Eng(oracle)
    inst = Instance(eng, B=6, n_items=4)
    out = afp25.decrypt_batch(eng, inst.g1, inst.tau_powers, inst.D, inst.f, inst.sk, inst.items)
    for t, item in enumerate(inst.items):
        assert (out[t] == inst.msgs[t]).all()                                   # round trip (afp25_bibe_test.go:51-108)
        assert (out[t] == inst.reference_shaped_decrypt(oracle, item)).all()
    # an identity outside the batch cannot decrypt (afp25_bibe_test.go:298-370)
    with pytest.raises(ValueError, match="identity not found"):
        afp25.decrypt_batch(eng, inst.g1, inst.tau_powers, inst.D, inst.f, inst.sk, [(12345, inst.items[0][1], inst.items[0][2])])


def test_duplicated_identity_is_an_error_as_in_the_reference(oracle):
    """bibe/afp25_bibe/afp25_bibe.go:371-381 removes every identity equal to id and errors unless exactly one was removed: a
    batch that lists an identity twice cannot be decrypted for it, although id is still a root of f."""
    eng = Eng(oracle)
    inst = Instance(eng, B=6, n_items=2)
    ids = list(inst.ids)
    out = afp25.decrypt_batch(eng, inst.g1, inst.tau_powers, inst.D, inst.f, inst.sk, inst.items, identities=ids)
    assert (out[0] == inst.msgs[0]).all()
    dup = ids + [inst.items[0][0]]
    with pytest.raises(ValueError, match="identity not found in identity list"):
        afp25.decrypt_batch(eng, inst.g1, inst.tau_powers, inst.D, inst.f, inst.sk, inst.items[:1], identities=dup)
    with pytest.raises(ValueError, match="identity not found in identity list"):
        afp25.decrypt_batch(eng, inst.g1, inst.tau_powers, inst.D, inst.f, inst.sk, inst.items[:1], identities=[i for i in ids if i != inst.items[0][0]])
