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
def golden():
    return np.load(os.path.join(GOLDEN, "reference_vectors.npz"))


@pytest.fixture(scope="session")
def golden_meta():
    import json
    with open(os.path.join(GOLDEN, "reference_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def meshes():
    return np.load(os.path.join(ROOT, "shapes", "meshes.npz"))


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from contexture_nerf_amd import _lib
    _lib.check(_lib.load().ctx_device_check())
    return torch.device("cuda:0")
