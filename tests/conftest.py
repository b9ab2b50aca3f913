import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "text-compression_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "hunit_vectors.json")) as f:
        return json.load(f)


def _have_gpu():
    """A device is present iff the kernel driver node exists and torch counts a device
    (device_count() does not initialise the GPU).  Deliberately NOT derived from the product:
    with a device present, a failing tc_ctx_create must fail the gpu tests, not skip them."""
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no MI355X on this host (gpu tests run on the GPU box with -m gpu)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
