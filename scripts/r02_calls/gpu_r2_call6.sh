#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_kernels.py tests/test_gpu_dist.py -m gpu -q -n 4 --timeout=600 2>&1 | tail -40 ) > gpurun_out/gpu_tests6.log; tail -40 gpurun_out/gpu_tests6.log
( timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 2>&1 | tail -1 | cut -c1-1200 ) | tee gpurun_out/bench6.log
