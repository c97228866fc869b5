import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the C oracle and (if hipcc is present and the .so is stale or missing) the HIP
    library once per session.  On the GPU box the prebuilt files travel with the snapshot."""
    from oracle import lbm_ref
    lbm_ref.build()
    from latticeboltzmannsimulations_amd import _lib
    if os.path.exists(_lib.HIPCC):
        _lib.build()
    yield
