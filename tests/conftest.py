import os
import sys

import pytest

# PyTorch bundles its own HIP runtime and RCCL images.  Loaded FIRST, libsph_hip.so binds to them (same sonames) and the
# process holds ONE runtime image; loaded after it, the process would map two (the cause of the same-process import crash
# of round 1, see tests/test_mesh_properties.py).  The product itself never needs torch.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure, oracle/). Built on demand with gcc."""
    from oracle import sph_oracle
    sph_oracle.build()
    return sph_oracle
