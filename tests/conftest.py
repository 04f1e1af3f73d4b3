import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liboracle.so), built on demand.  Test infrastructure only."""
    from oracle import oracle as O
    O.build()
    return O


def synth_image(seed, c, H, W):
    """SURVEY.md 8(d) pixel-domain generator."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((c, H, W))
    b = np.cumsum(np.cumsum(g, axis=1), axis=2)
    mn = b.min(axis=(1, 2), keepdims=True)
    mx = b.max(axis=(1, 2), keepdims=True)
    b = (b - mn) / (mx - mn)
    b = b + 0.02 * rng.standard_normal((c, H, W))
    return np.round(np.clip(b, 0, 1) * 255).astype(np.uint8) / 255


def synth_coeffs(seed, c, h, w, ll_h, ll_w, scale=3000.0):
    """SURVEY.md 8(d) coefficient-domain generator."""
    rng = np.random.default_rng(seed)
    i = np.arange(h)[:, None]
    j = np.arange(w)[None, :]
    t = np.maximum(0, np.ceil(np.log2(np.maximum((i + 1) / ll_h, (j + 1) / ll_w))))
    sc = scale * 2.0 ** (-1.3 * t)
    return np.trunc(rng.laplace(0, 1, (c, h, w)) * sc[None]).astype(np.int32)
