#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -q --timeout=600 2>&1 | tail -40 ) > gpurun_out/golden.log
( timeout 600 python __graft_entry__.py smoke 2>&1 | tail -15 ) > gpurun_out/smoke.log
( timeout 1200 python bench.py 2>&1 | tail -25 ) > gpurun_out/bench.log
cd /tmp && ( timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -5 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*stats*" | head; find gpurun_out/prof -type f -size +20M -delete
tail -30 gpurun_out/golden.log; cat gpurun_out/smoke.log; cat gpurun_out/bench.log; tail -3 gpurun_out/rocprof.log
