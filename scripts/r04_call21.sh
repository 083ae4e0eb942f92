#!/bin/bash
# GPU call 21: f32 tiled GEMM with operands three K steps ahead + 32-column tiles for few rows: strict-f32 tests + the f32 pass
mkdir -p gpurun_out
echo "== f32-heavy tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_golden.py -q -n 3 --timeout=800 --tb=short 2>&1 | tail -6
echo "== f32 pass"; ( timeout 400 python bench.py --dtype f32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 2>&1 | tail -1 ) | tee gpurun_out/r04_c21_bench_f32.json | cut -c1-300
