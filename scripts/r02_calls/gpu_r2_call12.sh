#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== dtw + loudness tests"; ( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -n 4 --timeout=300 -k "dtw or loudness" 2>&1 | tail -6 )
echo "== dtw timing"; ( timeout 100 python scripts/kernel_bench.py --only dtw 2>&1 | grep "W=" ) | tee gpurun_out/kb_dtw_gen3.txt
echo "== golden tests"; ( timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -q -n 4 --timeout=600 2>&1 | tail -4 )
echo "== align mode"; ( timeout 400 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 2>&1 | tail -1 ) | tee gpurun_out/bench12_align.json | cut -c1-2200
echo "== bench"; ( timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --phase-times 2>&1 | tail -1 ) | tee gpurun_out/bench12.json | cut -c1-1800
