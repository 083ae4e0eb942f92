"""The multi-GPU code path on the REAL backend with the hardware that is reachable: one rank.  `SWX_FORCE_DIST=1` makes
parallel.init_from_env() join a world of size 1 on the "nccl" backend (= RCCL on ROCm), so bench.py's RCCL broadcast of the
packed weight arena, the barrier, the max-over-ranks all-reduce and the result gather all execute on the device backend
instead of being skipped.  (N > 1 cannot be launched from this harness; the N = 2 logic is covered on CPU with gloo:
tests/test_parallel_cpu.py, tests/test_spans_cpu.py.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_single_rank_rccl_in_subprocess():
    env = dict(os.environ, SWX_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29600 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--model", "tiny.en", "--minutes", "1",
                        "--batch", "2", "--steps", "1", "--warmup", "0", "--tokens", "16", "--beam", "5", "--no-cpu-baseline",
                        "--no-roofline", "--no-f32"], capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["n_gpus_measured"] == 1 and out["value"] > 0
    assert out["config"]["per_rank"] == [dict(rank=0, segments=out["config"]["segments"], words=out["config"]["words"])]
