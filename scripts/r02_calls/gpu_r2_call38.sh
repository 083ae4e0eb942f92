#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== golden tests"; ( timeout 200 python -m pytest tests/test_gpu_golden.py -m gpu -q --timeout=300 2>&1 | tail -3 )
echo "== bench"; ( timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --phase-times 2>&1 | tail -1 ) | tee gpurun_out/bench38.json | cut -c1-1700
