import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("NIW_TEST_HOSTILE_MEMORY"):
        # diagnostic run (round 6): every torch.empty / resize comes filled with NaN (floats) or the largest integer, so that a kernel or
        # a wrapper that reads an output buffer before writing it fails a parity test instead of passing on benign recycled values:
        #     NIW_TEST_HOSTILE_MEMORY=1 python -m pytest tests -m gpu -q
        import torch
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
