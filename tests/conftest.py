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


_STREAM_DEFAULT = None


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
            # the residual-stream storage mode is process-wide (ta_set_stream_modes): every test starts from the default (bf16, or
            # whatever TA355_*_F32 asked for at load) whatever the previous test's ASRConfig.model_dtype left behind
            from tiny_audio_amd import ops
            global _STREAM_DEFAULT
            if _STREAM_DEFAULT is None:
                _STREAM_DEFAULT = ops.get_stream_modes()
            ops.set_stream_modes(**_STREAM_DEFAULT)
    yield
