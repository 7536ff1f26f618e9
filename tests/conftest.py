import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def load_golden(name):
    with gzip.open(os.path.join(GOLDEN, name), "rb") as f:
        return json.loads(f.read().decode())


@pytest.fixture(scope="session")
def acvp_keygen():
    return load_golden("acvp_keyGen.json.gz")


@pytest.fixture(scope="session")
def acvp_siggen():
    return load_golden("acvp_sigGen.json.gz")


@pytest.fixture(scope="session")
def acvp_sigver():
    return load_golden("acvp_sigVer.json.gz")


@pytest.fixture(scope="session")
def ref_hex():
    return load_golden("reference_hex_vectors.json.gz")


PSET = {"ML-DSA-44": 44, "ML-DSA-65": 65, "ML-DSA-87": 87}
