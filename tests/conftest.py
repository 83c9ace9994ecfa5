import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def tv():
    """torch.ops.torchvision with our library loaded."""
    import torch
    import vision_amd  # noqa: F401

    return torch.ops.torchvision


@pytest.fixture(scope="session")
def ref_loaded():
    """True when the REAL reference CPU kernels (oracle/_ref) are registered on the CPU key."""
    from oracle import oracle as O

    return O.load_reference()


@pytest.fixture(scope="session")
def need_ref(ref_loaded):
    if not ref_loaded:
        pytest.skip("oracle/_ref (compiled reference CPU kernels) not available")
    return True


@pytest.fixture(autouse=True)
def _seed():
    import torch

    torch.manual_seed(0)
    yield
