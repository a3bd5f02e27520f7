import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")
    # The CPU tier runs in a small VM whose vCPUs can be descheduled by the host: an 8-thread OpenMP team then spins at
    # every barrier and the suite goes from 10 s to 8 min.  The cases are sized for 1-2 threads (about 30 s).
    try:
        import torch
        if not torch.cuda.is_available():
            torch.set_num_threads(2)
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """GPU-marked tests must never silently pass on a box without a GPU."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu-marked test)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
