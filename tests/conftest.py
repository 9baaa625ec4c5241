import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: exhaustive checks (minutes of CPU)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.build()
    return o


@pytest.fixture(scope="session")
def synth():
    import lightloam_amd  # noqa: F401
    from lightloam_amd import synth as s
    return s


@pytest.fixture(scope="session")
def api():
    """The HIP C-ABI binding.  Never falls back: a missing library is a failure on a GPU box."""
    import lightloam_amd  # noqa: F401
    from lightloam_amd import api as a
    a.load_library()
    return a


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    neq = bits(a) != bits(b)
    if neq.any():
        idx = np.argwhere(neq)[:5]
        raise AssertionError(f"{what}: {int(neq.sum())} of {neq.size} f32 values differ bitwise; first at {idx.tolist()}: "
                             f"{a[tuple(idx[0])]!r} vs {b[tuple(idx[0])]!r}")


ORG_PATHS = {"tiles": "64", "walk": "0"}


def set_org_path(name):
    """Which organise kernels the contexts created from now on use for small calls: "tiles" = the tile-parallel path of
    calls of at most 64 scans (the library's default), "walk" = k_organize (one workgroup per scan) for every call."""
    os.environ["LIGHTLOAM_ORG_SMALL"] = ORG_PATHS[name]
