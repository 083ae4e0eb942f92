#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
export SWX_DEC_MIN_ROWS=1
echo "== grid barrier microbench"; timeout 120 scripts/micro/gbar 2>&1 | tee gpurun_out/grid_barrier.txt
echo "== model tests"; ( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -n 4 --timeout=600 2>&1 | tail -25 ) > gpurun_out/gpu_tests8.log; tail -25 gpurun_out/gpu_tests8.log
echo "== align mode"; ( timeout 400 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 2>&1 | tail -1 ) | tee gpurun_out/bench8_align.json | cut -c1-1800
echo "== align mode, per-op small pass"; ( SWX_FLAGS=340 timeout 400 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 ) | cut -c1-300
