import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are SKIPPED (not failed) on a machine without an MI355X or without the built library, so a
    plain `pytest tests` works everywhere; `-m gpu` on the GPU box runs them."""
    import torch
    lib = os.path.join(REPO, "holo_diffusion_amd", "libholo_mi355x.so")
    if torch.cuda.is_available() and os.path.isfile(lib):
        return
    why = "no HIP device" if not torch.cuda.is_available() else "libholo_mi355x.so is not built"
    skip = pytest.mark.skip(reason=f"gpu test: {why}")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
