import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    import fwapi

    fwapi.build_oracle()

# One process, one copy of the HIP runtime: PyTorch ships its own libamdhip64, libfwgpu links against /opt/rocm's.  Whoever
# is loaded first wins the SONAME; loading torch first lets libfwgpu bind to the copy torch already mapped (the other
# order leaves torch without a usable device: "No HIP GPUs are available").  Tests that hand torch tensors to the C ABI
# need both, so torch goes first here.
try:
    import torch  # noqa: F401,E402
except ImportError:  # CPU tier without torch: nothing to order
    pass
