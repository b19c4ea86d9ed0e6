import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    """Counting devices does not initialise the GPU in this process (is_available() would): test_dist_gpu.py starts
    its child ranks from a process that has not touched the device."""
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # the multi-process GPU tests run first: they spawn fresh child ranks and must do so before any other test has
    # initialised the GPU in this (the parent) process
    items.sort(key=lambda it: 0 if "test_dist_gpu" in it.nodeid else 1)
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
