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
    """The product package.  The C-ABI library is normally built in-tree by
    __graft_entry__.build(); on a fresh checkout build it here (hipcc
    cross-compiles for gfx950 without a GPU)."""
    import grad_traj_optimization_amd as g
    if not os.path.exists(g.library_path()):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "grad_traj_optimization_amd", "csrc"), "-s", "all"])
    g.load_library()
    return g
