import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    """build the C-ABI library and the oracle if they are missing (CPU-only, hipcc cross-compiles)"""
    import __graft_entry__ as ge
    ge.build()
    return True
