import os
import sys

import pytest

# Load torch (and the HIP runtime it bundles) BEFORE libsplat_hip.so pulls in /opt/rocm's: the tests that use
# torch for device memory / torch.distributed otherwise find "No HIP GPUs" when a test file is run on its own.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is plumbing for a few tests only
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O
