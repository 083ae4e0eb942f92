#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== model + golden + largev3 tests"; ( timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_largev3.py -m gpu -q -n 4 --timeout=600 2>&1 | tail -4 )
echo "== bench"; ( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
echo "== align"; ( timeout 600 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
