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


def pytest_collection_modifyitems(config, items):
    # a gpu-marked test on a box without a GPU is an error in how the suite was invoked, but the
    # CPU-only driver run uses -m "not gpu", so only guard against accidental unmarked collection
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return x.to(dtype) if dtype is not None else x
