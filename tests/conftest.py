import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_point_ops():
    return np.load(os.path.join(GOLDEN, "point_ops.npz"))


@pytest.fixture(scope="session")
def golden_model():
    return np.load(os.path.join(GOLDEN, "model.npz"))


@pytest.fixture(scope="session")
def golden_loss():
    return np.load(os.path.join(GOLDEN, "loss.npz"))


@pytest.fixture(scope="session")
def golden_eval():
    return np.load(os.path.join(GOLDEN, "eval.npz"))


@pytest.fixture(scope="session")
def golden_data():
    """tests/golden/data.npz: the reference's data-pipeline functions on seeded clouds (make_golden_data.py)."""
    return np.load(os.path.join(GOLDEN, "data.npz"))


@pytest.fixture(scope="session")
def golden_data2():
    """tests/golden/data2.npz: the reference's double-cut CADDataset item and BuildingDataset item (make_golden_data2.py)."""
    return np.load(os.path.join(GOLDEN, "data2.npz"))


def golden_cloud(seed, M):
    """The raw cloud of a data2.npz case, regenerated from its seed exactly as make_golden_data2.py made it."""
    rng = np.random.default_rng(50_000 + int(seed))
    return (rng.random((int(M), 3), dtype=np.float32) - np.float32(0.5)).astype(np.float32)
