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
    config.addinivalue_line("markers", "perf: wall-clock assertions on a GPU (run with -m perf on a quiet box; in neither the CPU nor the -m gpu set)")


def pytest_collection_modifyitems(config, items):
    # A live timing assertion fails on noise one day and, under the driver's `-x`, hides every test behind it
    # (VERDICT r5 weak 12 / ADVICE r5): such tests run only when asked for by name of their marker.
    if "perf" in (config.option.markexpr or ""):
        return
    skip = pytest.mark.skip(reason="wall-clock assertion: run with -m perf on a quiet GPU box")
    for it in items:
        if "perf" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O
