import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/liborc.so) — the checker, never the thing under test."""
    from tests import _orc
    return _orc.Oracle()


@pytest.fixture(scope="session")
def lib():
    """libstorm_hip.so through ctypes (the product's C-ABI)."""
    import stormbitmaps_amd as sb
    return sb.load()


@pytest.fixture(scope="session")
def hip_ctx():
    import stormbitmaps_amd as sb
    ctx = sb.HipContext(0)
    yield ctx
    ctx.close()
