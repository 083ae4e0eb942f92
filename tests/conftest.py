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
    # first-ever hardware runs of new device code (non-strict xfail, own subprocess) go last: whatever they do to the
    # GPU, the validated parity tests have already run
    pending = [it for it in items if it.get_closest_marker("xfail") is not None and "gpu" in it.keywords]
    if pending:
        items[:] = [it for it in items if it not in pending] + pending
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
