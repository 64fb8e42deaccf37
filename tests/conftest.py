import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "face-recognition_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def weights_f32():
    """Synthetic VGG-Face weights (float32, KRSC) shared by the whole session."""
    from frhot.weights import synthetic_weights
    return synthetic_weights(seed=0xFACE)


@pytest.fixture(scope="session")
def weights_q(weights_f32):
    """The same weights as the library stores them (fp16-rounded matrices)."""
    from frhot.weights import quantize_fp16
    return quantize_fp16(weights_f32)


@pytest.fixture(scope="session")
def gpu_ctx(weights_f32):
    """A context on cuda:0 with the synthetic model loaded (GPU tests only)."""
    from frhot.engine import Context
    from frhot.weights import pack_blob
    ctx = Context(device=0, max_batch=16)
    ctx.load_model(pack_blob(weights_f32), input_scale=1.0 / 16.0)
    yield ctx
    ctx.close()


def cosine_rows(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
