import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_bytes(name: str) -> bytes:
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.build()
    return o


@pytest.fixture(scope="session")
def h5():
    return golden_bytes("NC_007362.1.txt")


@pytest.fixture(scope="session")
def h1():
    return golden_bytes("NC_026433.1.txt")


@pytest.fixture(scope="session")
def cy():
    return golden_bytes("CY137594.txt")
