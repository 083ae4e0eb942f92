#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== flash timing"; ( timeout 100 python scripts/kernel_bench.py --only flash 2>&1 | tail -3 ) | tee gpurun_out/kb_flash2b.txt
echo "== kernel + model tests"; ( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_largev3.py -m gpu -q -n 4 --timeout=600 2>&1 | tail -4 )
echo "== bench"; ( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
