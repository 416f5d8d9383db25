import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle_lib

    return oracle_lib.load()


@pytest.fixture(scope="session")
def ctx():
    """One device context for the whole GPU session (one process, one GPU)."""
    from align3d_amd import Context

    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def diag_ctx():
    """A context on the DIAGNOSTICS build of the library (libalign3d_hip_diag.so, -DA3D_DIAGNOSTICS): the environment
    knobs, the exact-arithmetic cross-check kernel, the kernel variants that were measured slower and the cross-check
    paths (rocPRIM sort, host kd-tree build, last-block hand-off) exist only there.  Tests that need one of those use
    this fixture; everything else runs on the product library."""
    from align3d_amd import Context, _abi

    c = Context(0, library=_abi.DIAG_LIB_PATH)
    yield c
    c.close()
