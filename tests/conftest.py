import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _limit_threads():
    # the CPU oracle runs inside the GPU tests; xdist workers x 64 default torch threads thrash the host
    if not os.environ.get("PYTEST_XDIST_WORKER"):
        return
    try:
        import torch
        torch.set_num_threads(max(1, min(8, len(os.sched_getaffinity(0)))))
    except Exception:
        pass


def pytest_configure(config):
    _limit_threads()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver)")
    config.addinivalue_line("markers", "slow: long CPU test")


def pytest_collection_modifyitems(config, items):
    # checks that run in a subprocess of their own (hw_checks/, inner pytest runs) go last
    pending = [it for it in items if "gpu" in it.keywords and "in_subprocess" in it.name]
    if pending:
        items[:] = [it for it in items if it not in pending] + pending
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
