import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as G

    return G.load_package()


@pytest.fixture(scope="session")
def ctx(pkg):
    """One device context for the whole GPU session.  Fails loudly when the HIP library or the GPU is missing."""
    c = pkg.Context(0)
    yield c
    c.close()
