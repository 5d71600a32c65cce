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


def shipped(ctx, key, values):
    """The values of a kernel-selecting option that THIS build of the library carries: the shipped
    libstorm_hip.so keeps one form per output kind plus one independent operand path; the slower alternative
    forms (32x32x64 / wide / persistent strips, the other output kernels, stripbits_kernel, bitwave_kernel) live in the tools
    build (`make probes`, STORM_HIP_LIB) and the shipped library refuses the options that select them."""
    probes = ctx.get_option("probes_build") == 1
    only_probes = {"variant": {5}, "k2_shape": {32}, "k2_persistent": {1}, "k2_tile_shape": {1, 16},
                   "k2_strip_operands": {1, 3}}.get(key, set())
    return tuple(v for v in values if probes or v not in only_probes)
