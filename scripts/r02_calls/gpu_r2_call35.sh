#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8; do
  timeout 200 python -m pytest tests/test_gpu_golden.py -m gpu -q -x -k "temperature_ladder" -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  |assert" | head -12
done 2>&1 | tee gpurun_out/ladder_repeat.txt
