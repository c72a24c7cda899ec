"""pytest configuration: markers + path setup.

``-m "not gpu"``: oracle vs golden vectors, host logic, C-ABI symbol checks,
gloo world_size-2 data-parallel tests -- no GPU required.
``-m gpu``: HIP path vs the oracle through the C-ABI on a real MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver via gpurun)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(autouse=True)
def _poison_free_gpu_memory(request):
    """-m gpu tests: fill the allocator's free pool with NaN bit patterns before every test, so that a kernel reading memory
    nobody wrote (workspace padding rows, an unwritten gradient slice ...) fails loudly instead of depending on what ran before."""
    if "gpu" in request.keywords:
        import torch
        if torch.cuda.is_available():
            blocks = [torch.full((64 * 1024 * 1024,), float("nan"), device="cuda") for _ in range(4)]    # 4 x 256 MB
            small = [torch.full((n,), float("nan"), device="cuda") for n in (256, 4096, 65536, 1 << 20) for _ in range(8)]
            del blocks, small
    yield
