"""BASELINE config 4 semantics (cpabe/bsw07 decrypt) through the host planner gopairingbasedcryptography_amd/bsw07.py:
the folded one-multi-pairing decryption must return the encrypted message AND be bit-identical to the reference-shaped
evaluation (n full pairings, GT.Div, GT.Exp, GT.Mul) — SURVEY §8a-3's restructuring claim.  CPU version on the oracle
engine here; the same flow runs on the GPU engine in test_gpu_parity.py."""
import numpy as np
import pytest

import bn254_py as o
from bsw07_fixture import Instance, example_tree
from gopairingbasedcryptography_amd import bsw07


class OracleEngine:
    def __init__(self, oracle):
        self.o = oracle

    def _k(self, ks):
        if isinstance(ks, (list, tuple)):
            return np.frombuffer(b"".join(o.scalar_to_bytes(int(k) % o.R) for k in ks), dtype=np.uint8)
        return ks

    def pair_batch(self, P, Q): return self.o.pair_batch(P, Q)
    def multi_pair(self, P, Q, off): return self.o.multi_pair(P, Q, off, threads=4)
    def g1_scalar_mul(self, b, k): return self.o.g1_scalar_mul(b, self._k(k))
    def g2_scalar_mul(self, b, k): return self.o.g2_scalar_mul(b, self._k(k))
    def g2_sum(self, p): return self.o.g2_sum(p)
    def gt_mul(self, a, b): return self.o.gt_mul(a, b)
    def gt_exp(self, x, k): return self.o.gt_exp(x, self._k(k))


def test_lagrange_and_plan():
    assert bsw07.lagrange_at_zero(1, [1, 2]) == 2 and bsw07.lagrange_at_zero(2, [1, 2]) == o.R - 1
    tree = example_tree()
    bsw07.assign_leaf_ids(tree)
    assert bsw07.decrypt_plan(tree, {44}) is None
    plan = bsw07.decrypt_plan(tree, {11, 22, 33})
    # root 2-of-4 uses children 1 (leaf 11) and 2 (the 2-of-2 gate): Delta_1 = 2, Delta_2 = -1; inside the gate 2 and -1
    assert plan == {1: (11, 2), 2: (22, (o.R - 1) * 2 % o.R), 3: (33, 1)}


def test_folded_decrypt_matches_reference_shape(oracle):
    eng = OracleEngine(oracle)
    inst = Instance(eng, example_tree(), user_attrs=[11, 22, 33, 99], n_ct=3)
    plan = bsw07.decrypt_plan(inst.tree, inst.user_attrs)
    folded = bsw07.fold_key(eng, plan, inst.dj, inst.dj_prime)
    out = bsw07.decrypt_batch(eng, folded, inst.D, inst.cts, Instance.neg_g1)
    for t, ct in enumerate(inst.cts):
        assert (out[t] == inst.msgs[t]).all()                              # round trip (bsw07_cpabe_test.go:12-80)
        assert (out[t] == inst.reference_shaped_decrypt(oracle, ct)).all()  # bit-identical to the reference's evaluation order


def test_unsatisfied_policy(oracle):
    eng = OracleEngine(oracle)
    inst = Instance(eng, example_tree(), user_attrs=[22, 44 + 1], n_ct=1)
    assert bsw07.decrypt_plan(inst.tree, inst.user_attrs) is None          # bsw07_cpabe_test.go:83-141: decrypt must fail
    assert inst.reference_shaped_decrypt(oracle, inst.cts[0]) is None
