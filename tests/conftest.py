import os
import sys

import pytest

# The CPU oracle's matmuls must not depend on the machine's load: with dynamic thread adjustment a busy host (xdist workers
# sharing a CPU quota) changes the BLAS / OpenMP partition from call to call, i.e. the last bits of the logits, and a token
# comparison on random weights can then flip at a near-tie (seen once in ~6 runs of the GPU suite under -n 4).
os.environ.setdefault("MKL_DYNAMIC", "FALSE")
os.environ.setdefault("OMP_DYNAMIC", "FALSE")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_quota() -> int:
    """CPUs this process may actually use: the cgroup quota when there is one (the MI355X boxes show 256 hardware threads under
    a 16-CPU quota), else the affinity mask"""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def _limit_threads():
    # the CPU oracle runs inside the GPU tests.  torch sizes its intra-op pool by the visible hardware threads; more OpenMP
    # workers than the container's CPU quota get the whole process throttled (they busy-wait between parallel regions), and
    # xdist workers multiply it
    try:
        import torch
        n = _cpu_quota()
        if os.environ.get("PYTEST_XDIST_WORKER"):
            # the workers SHARE the quota: 4 workers x 4 threads on 8 cores ran a 10-s oracle test in 114 s (OpenMP workers spin
            # between parallel regions), 4 x 2 threads in 13 s
            workers = int(os.environ.get("PYTEST_XDIST_WORKER_COUNT", "2") or 2)
            n = max(1, n // max(1, workers))
        if torch.get_num_threads() > n:
            torch.set_num_threads(n)
    except Exception:
        pass


def subprocess_env():
    """environment for the hardware checks that run in a child process (tests/hw_checks/*.py): torch sizes its intra-op pool by the
    VISIBLE hardware threads (256 on the MI355X boxes) and the CPU oracle inside those scripts then runs 256 OpenMP workers under a
    16-CPU quota -- score_qk_check.py took 145 s of the serial GPU suite that way (17 s of CPU work).  The children get the quota."""
    n = str(_cpu_quota())
    return dict(os.environ, OMP_NUM_THREADS=n, MKL_NUM_THREADS=n)


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
