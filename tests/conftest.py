import os
import random
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def set_rng_seed(seed: int) -> None:
    """Same helper as the reference's tests/test_utils.py:62-65."""
    torch.manual_seed(seed)
    random.seed(seed)


@pytest.fixture
def golden():
    import numpy as np

    def load(name):
        return np.load(GOLDEN / name, allow_pickle=False)

    return load
