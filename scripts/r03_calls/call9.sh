#!/bin/bash
# round 3, call 9 (= call 5 on the final code): the whole GPU suite on the final code, per-kernel durations (eager, prefetch on / off), then the measurement
# set (PMC traffic, rocprofv3 summary under graph replay, the default bench line with roofline / strict f32 / CPU baseline)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c9
mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
stamp() { echo "=== $1 (t+$(( $(date +%s) - t0 ))s)"; }
stamp "full GPU suite"
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
stamp "smoke"
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
stamp "per-kernel durations, eager launches, prefetch on / off"
scripts/rocprof_kernels.sh r03_eager_pf_on python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --debug-flags 16384
scripts/rocprof_kernels.sh r03_eager_pf_off python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --debug-flags 49152
head -12 gpurun_out/r03_eager_pf_on_kernels.csv | cut -c1-150
head -12 gpurun_out/r03_eager_pf_off_kernels.csv | cut -c1-150
stamp "measurement set"
SKIP_SUITE=1 bash scripts/gpu_final.sh r03final > $O/gpu_final.log 2>&1
tail -3 $O/gpu_final.log | cut -c1-2500
stamp "done"
