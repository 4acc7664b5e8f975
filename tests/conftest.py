import os
import sys

import pytest

try:  # one HIP runtime per process: torch's bundled runtime must be loaded before libhelfem_amd.so
    import torch  # noqa: F401
except Exception:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_libs():
    """Build (incrementally) the product and oracle libraries once per session."""
    from helfem_amd import build
    lib = os.path.join(ROOT, "helfem_amd", "lib", "libhelfem_amd.so")
    if not os.path.exists(lib):
        build.build_product(verbose=False)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        build.build_oracle(verbose=False)
    return True
