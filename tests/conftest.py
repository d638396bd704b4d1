import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stanford-ctc_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must fail loudly on a GPU box without the HIP library; on a box
    # without any GPU they are deselected by `-m "not gpu"` (the driver's CPU run).
    pass


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(autouse=True)
def _numpy_errstate():
    # importing the ctc_fast surface flips np.seterr globally (reference a3);
    # keep tests independent of import order
    old = np.geterr()
    yield
    np.seterr(**old)


@pytest.fixture(autouse=True)
def _seeded_rngs():
    # every test starts from the same global generator states (NumPy's legacy generator, torch CPU and
    # device): a tolerance that holds once holds on every box
    np.random.seed(20240927)
    try:
        import torch
        torch.manual_seed(20240927)
    except ImportError:
        pass
    yield
