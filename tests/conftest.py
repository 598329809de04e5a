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
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def load_golden(name):
    """-> (arrays dict as torch tensors, state dict under 'sd/', grads under 'grad/')."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    arrs = {k: torch.from_numpy(np.asarray(z[k])) if z[k].dtype.kind in "fiub" else z[k] for k in z.files}
    sd = {k[3:]: v for k, v in arrs.items() if k.startswith("sd/")}
    grads = {k[5:]: v for k, v in arrs.items() if k.startswith("grad/")}
    return arrs, sd, grads


@pytest.fixture(scope="session")
def golden():
    return load_golden
