import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hip_lib():
    """libfloria_hip.so built in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    from floria_amd import lib
    lib.build()
    return lib


@pytest.fixture(scope="session")
def gpu_ctx(hip_lib):
    ctx = hip_lib.FloriaHip(0)        # raises (no CPU fallback) when no device is usable
    yield ctx
    ctx.close()
