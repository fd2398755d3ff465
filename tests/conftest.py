import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """On a GPU box bring torch's device context up before the first dycore handle: torch
    (band buffers, torch.distributed) and libdycore_hip.so share one HIP runtime in the process,
    and torch's lazy initialisation failed ("No HIP GPUs are available") when it came late in a
    long session."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device='cuda')
    except Exception:
        pass
