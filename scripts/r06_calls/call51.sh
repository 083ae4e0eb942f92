#!/bin/bash
# usage: gpurun --timeout 900 -- 'bash scripts/r06_calls/call51.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 800 python -m pytest tests/test_gpu_model_families.py -q -m gpu -k "end_to_end" --durations=4 -p no:cacheprovider 2>&1 | tail -40 ) > gpurun_out/r06_c51_e2e.log
tail -30 gpurun_out/r06_c51_e2e.log
