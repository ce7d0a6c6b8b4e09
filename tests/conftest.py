import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(REPO, "tests", "golden", "toad_golden.npz")
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope="session")
def golden_cases(golden):
    names = sorted({k.split("/")[0] for k in golden.files if "/" in k and not k.startswith("api/")})
    return names


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from toad_amd import _lib
    _lib.load()          # fail loudly if the extension is missing on a GPU box
    return torch.device("cuda:0")
