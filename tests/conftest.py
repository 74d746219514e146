import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(HERE, "golden", "bn254_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def lsa():
    """The HIP library, initialised on cuda:0.  Fails loudly if it is missing."""
    import legosnark_amd
    legosnark_amd.init(0)
    return legosnark_amd


# ---- decoding helpers for the golden JSON (canonical hex -> python ints) -------------
def g1_dec(p):
    return None if p == "inf" else (int(p[0], 16), int(p[1], 16))


def g2_dec(p):
    if p == "inf":
        return None
    return ((int(p[0][0], 16), int(p[0][1], 16)), (int(p[1][0], 16), int(p[1][1], 16)))


def f12_dec(f):
    return [(int(c[0], 16), int(c[1], 16)) for c in f]
