import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# errors the HIP runtime / RCCL report are printed (both are silent by default); must be in the environment before the runtime starts
# (pytest.ini has the story of the silent abort this is here for)
os.environ.setdefault("AMD_LOG_LEVEL", "1")
os.environ.setdefault("NCCL_DEBUG", "WARN")

# the lane tests run in this process: demon_amd.lanes exports GPU_MAX_HW_QUEUES when it is imported, which must happen before the
# first HIP call (bench.py imports it at its top for the same reason)
import demon_amd.lanes  # noqa: E402,F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_l1(a, b):
    """relative L1 error used by the parity gate (BASELINE.json: <= 1e-3)"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-30))


def make_inputs(n, height=192, width=256, seed=0):
    """Synthetic pair batch of SURVEY.md section 8(d): uniform [-0.5,0.5), image2_2 = 4x4 box mean of image 2."""
    rng = np.random.default_rng(seed)
    pair = rng.random((n, 6, height, width), dtype=np.float32) - np.float32(0.5)
    img2_2 = pair[:, 3:6].reshape(n, 3, height // 4, 4, width // 4, 4).mean(axis=(3, 5)).astype(np.float32)
    return pair, img2_2


@pytest.fixture(scope="session")
def synth_weights():
    from demon_amd import weights
    return weights.synthetic_weights(seed=1)


@pytest.fixture(scope="session")
def gpu_ctx(synth_weights):
    """A DemonContext with synthetic weights on cuda:0; fails loudly (no fallback) when the HIP path is absent."""
    from demon_amd import DemonContext
    guard = os.environ.pop("DEMON_POISON_GUARD", None)      # (tests/test_poison_gpu.py sets it per test: the shared context stays a plain one)
    try:
        ctx = DemonContext(device=0, max_batch=4, height=192, width=256)
    finally:
        if guard is not None:
            os.environ["DEMON_POISON_GUARD"] = guard
    ctx.set_weights(synth_weights)
    yield ctx
    ctx.close()
