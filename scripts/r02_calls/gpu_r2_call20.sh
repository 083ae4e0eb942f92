#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python scripts/micro/encode_b1.py 2>&1 | grep "B=" | tee gpurun_out/encode_b1.txt
echo "== cross-kv / model tests"; ( timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -n 4 --timeout=600 2>&1 | tail -5 )
echo "== align"; ( timeout 600 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
echo "== bench"; ( timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
