import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the path-forcing registry is test infrastructure (libgdf_testhook.so, csrc/testhook.cpp): the Python binding loads it in front of
# libgdf.so only when asked to -- the shipped library exports no switch of its own
os.environ["LIBGDF_AMD_TESTHOOK"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _seed():
    # the reference seeds its python tests with 0xabcdef (python/tests/conftest.py:14-20, utils.py:31-33)
    np.random.seed(0xabcdef)
    yield


@pytest.fixture(scope="session")
def gdf():
    """The GPU-side package.  Imports torch first so one HIP runtime serves torch and libgdf.so."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    import libgdf_amd
    return libgdf_amd


@pytest.fixture
def force_path(gdf):
    """Force one of the library's alternative code paths for the duration of a test: force_path("GDF_JK_NO_SPEC", "1").
    The shipped libgdf.so reads no environment variable (csrc/lab.h); the parity tests that run one request through two
    code paths select the second one through gdf_amd_debug_force of the test-hook library (libgdf_testhook.so), and this fixture clears every
    name it set when the test ends."""
    names = []

    def force(name, value="1"):
        gdf.libgdf.gdf_amd_debug_force(name.encode(), None if value is None else str(value).encode())
        if value is not None:
            names.append(name)

    yield force
    for name in names:
        gdf.libgdf.gdf_amd_debug_force(name.encode(), None)
