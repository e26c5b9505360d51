import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build the C-ABI library (hipcc cross-compiles
    # without a GPU) and the oracle before the test modules import them
    if not (os.path.exists(os.path.join(ROOT, "atlas_amd", "lib", "libatlas_amd.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))):
        import __graft_entry__ as ge
        ge.build()


@pytest.fixture(scope="session")
def lib_built():
    """build the C-ABI library and the oracle if they are missing (CPU-only, hipcc cross-compiles)"""
    import __graft_entry__ as ge
    ge.build()
    return True
