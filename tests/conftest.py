import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "codename-rvc-fork-3_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rms(x):
    x = np.asarray(x, dtype=np.float64)
    return float(np.sqrt(np.mean(x * x)))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def ref_inputs():
    """The reference's own realistic Synthesizer inputs (logs/reference/ref_*.npy, train.py:839-842)."""
    return (np.load(os.path.join(GOLDEN, "ref_feats.npy")), np.load(os.path.join(GOLDEN, "ref_f0c.npy")),
            np.load(os.path.join(GOLDEN, "ref_f0f.npy")))
