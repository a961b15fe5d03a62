import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tdtk():
    """The product package (directory name starts with a digit -> importlib)."""
    capi = importlib.import_module("3dtk_amd._capi")
    if not os.path.exists(os.path.join(ROOT, "3dtk_amd", "lib3dtk_hip.so")):
        capi.build_extension()
    return importlib.import_module("3dtk_amd")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (checker only)."""
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def gpu(tdtk):
    if tdtk.device_count() < 1:
        pytest.fail("no HIP device visible: the -m gpu tier must run on the GPU box "
                    "(the product has no CPU fallback)")
    return 0


@pytest.fixture
def lab(tdtk):
    """The LAB library for this test (3dtk_amd/lib3dtk_hip_lab.so: the product's sources with -DTDTK_LAB -- + the kernels
    and policies that were built, measured and lost, and the environment switches that select them).  Tests that compare
    such a variant with the product path run entirely inside it (its default path is the product's code); the product
    library holds none of those kernels and reads none of those switches.  test_lab_library_default_path_is_the_products
    ties the two together."""
    capi = importlib.import_module("3dtk_amd._capi")
    if not os.path.exists(os.path.join(ROOT, "3dtk_amd", "lib3dtk_hip_lab.so")):
        capi.build_extension()
    with capi.library("lab"):
        yield True
