#!/bin/bash
# usage: gpurun --timeout 900 -- 'bash scripts/r06_calls/call50.sh'
# the receiving-rank path of an N > 1 run (arena delivered from outside + mark_weights_loaded), emulated in one process: tests/test_gpu_dist.py
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 800 python -m pytest tests/test_gpu_dist.py -q -m gpu --durations=6 -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/r06_c50_dist.log
tail -25 gpurun_out/r06_c50_dist.log
