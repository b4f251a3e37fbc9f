"""Shared pytest configuration: markers, import paths and golden-vector loading."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vllm-backend-public_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box only)")


def load_npz(name):
    """Load a golden .npz; arrays stored as raw bf16 words (key suffix __bf16)
    are decoded to float32."""
    z = np.load(os.path.join(GOLDEN, name))
    out = {}
    for k in z.files:
        if k.endswith("__bf16"):
            out[k[:-6]] = (z[k].astype(np.uint32) << 16).view(np.float32)
        else:
            out[k] = z[k]
    return out


@pytest.fixture(scope="session")
def ops_golden():
    return load_npz("ops_hf.npz")
