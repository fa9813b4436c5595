import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (arrays dict of torch tensors, weights dict keyed without the 'w.' prefix)"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    arrs = {k: torch.from_numpy(z[k]) for k in z.files}
    return arrs


def split_prefix(arrs, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in arrs.items() if k.startswith(prefix)}


def rel_err(a, b):
    a = a.double()
    b = b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_rel(a, b):
    """max |a-b| / max|b| -- a scale-relative max-norm error."""
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
