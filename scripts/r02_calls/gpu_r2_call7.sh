#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_kernels.py -m gpu -q -n 4 --timeout=600 2>&1 | tail -12 ) > gpurun_out/gpu_tests7.log; tail -12 gpurun_out/gpu_tests7.log
for cfg in "SWX_DEC_MIN_ROWS=48" "SWX_DEC_MIN_ROWS=1"; do
  echo "== sequential $cfg"; ( env $cfg timeout 400 python bench.py --sequential --minutes 3 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-700 )
  echo "== batch 4 $cfg"; ( env $cfg timeout 400 python bench.py --batch 4 --minutes 4 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-400 )
  echo "== batch 8 $cfg"; ( env $cfg timeout 400 python bench.py --batch 8 --minutes 4 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-400 )
done
