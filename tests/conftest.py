import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "audio-formats_amd"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """The HIP path or nothing: skip only when no GPU is visible, never fall back to a CPU path."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import afgpu
    afgpu.lib()          # raises loudly if libafg_hip.so is missing
    return torch.device("cuda:0")
