import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# One HIP runtime per process: torch's bundled copy has to be loaded before libcpuvox_gpu.so pulls in /opt/rocm's (see
# cpuvox_amd.gpu._load_torch_hip_runtime_first); tests hand torch tensors to the C ABI, so make the order explicit here instead of
# depending on which test file happens to be collected first.
try:
    import torch  # noqa: F401,E402
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The product libraries must exist in-tree (they travel to the GPU box); build them if this is a fresh checkout."""
    import __graft_entry__ as g

    g.ensure_built()
    yield
