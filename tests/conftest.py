import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    """Builds (if needed) and loads the C oracle. Test infrastructure only."""
    import subprocess
    import pyoracle
    if not os.path.exists(pyoracle.oracle_lib_path()):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    return pyoracle.CEC()


@pytest.fixture(scope="session")
def gpu():
    """One library context on cuda:0; fails loudly if the HIP library or the GPU is missing."""
    import bulletproofspp_amd as b
    ctx = b.Bppp(0)
    yield ctx
    ctx.close()
