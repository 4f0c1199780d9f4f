import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def hx(s):
    """hex string -> uint8 array"""
    return np.frombuffer(bytes.fromhex(s), dtype=np.uint8)


def cat(hexes):
    return np.concatenate([hx(h) for h in hexes]) if len(hexes) else np.zeros(0, dtype=np.uint8)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


def eip197_pairs(words):
    """EIP-197 words (hex, big-endian) -> ([(x, y)], [((x.re, x.im), (y.re, y.im))]) as Python integers (tests/golden/eip197_pairing.json)"""
    w = [int(h, 16) for h in words]
    ps = [(w[6 * k], w[6 * k + 1]) for k in range(len(w) // 6)]
    qs = [((w[6 * k + 3], w[6 * k + 2]), (w[6 * k + 5], w[6 * k + 4])) for k in range(len(w) // 6)]
    return ps, qs
