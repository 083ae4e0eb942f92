#!/bin/bash
# usage: gpurun --timeout 1800 -- 'bash scripts/r06_calls/call49.sh'
# the serial GPU suite as the driver runs it, with tests/test_gpu_model_families.py in it (small.en / medium / large-v3-turbo), + smoke
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -30 ) > gpurun_out/r06_final14_gpu_suite.log; tail -4 gpurun_out/r06_final14_gpu_suite.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/r06_final14_smoke.log; cat gpurun_out/r06_final14_smoke.log
