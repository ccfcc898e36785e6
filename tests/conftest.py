import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def stable_seed(*parts) -> int:
    """A seed that is the same in every process (str hashes are salted per process; this is a CRC of the repr)."""
    import zlib

    return zlib.crc32(repr(parts).encode())


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


@pytest.fixture
def debug():
    """Kernel-selection overrides of the context on device 0 (zsw_debug_set); restored to the defaults after the test."""
    import zoe_amd
    from zoe_amd import _lib

    ctx = zoe_amd.SwContext.get(0)

    class _Debug:
        SCORE_V1, NO_TILES, NO_W32, NO_WIDE = _lib.DEBUG_SCORE_V1, _lib.DEBUG_NO_TILES, _lib.DEBUG_NO_W32, _lib.DEBUG_NO_WIDE
        NO_SIDE_STREAMS, NO_PIPELINE, ALIGN_NO_PACKED = _lib.DEBUG_NO_SIDE_STREAMS, _lib.DEBUG_NO_PIPELINE, _lib.DEBUG_ALIGN_NO_PACKED

        @staticmethod
        def set(flags=0):
            ctx.debug_set(flags)

    yield _Debug
    ctx.debug_set(0)
