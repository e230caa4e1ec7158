import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU restatement (test infrastructure), compiled on demand."""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gtop():
    """The product package; the C-ABI library must already be built in-tree."""
    import grad_traj_optimization_amd as g
    g.load_library()
    return g
