import os
import sys

import pytest

# The reference side of the GPU tests (oracle / plain-PyTorch ops) runs torch's MIOpen convolutions; on a fresh box MIOpen's
# exhaustive per-shape solver search costs minutes.  Immediate-mode heuristics are enough for a checker (the product path does
# not use MIOpen at all).
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped (not failed) on a box without a GPU, so a plain `pytest tests` works in the build container"""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="needs a GPU (pytest -m gpu on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
