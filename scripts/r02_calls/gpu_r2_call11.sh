#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== dtw tests"; ( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -n 4 --timeout=300 -k "dtw" 2>&1 | tail -12 )
for cfg in "SWX_DTW_GEN2=0" "SWX_DTW_GEN2=1"; do
  echo "== dtw timing $cfg"; ( env $cfg timeout 100 python scripts/kernel_bench.py --only dtw 2>&1 | grep "W=" )
done | tee gpurun_out/kb_dtw_gen3.txt
echo "== golden + model tests"; ( timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_model.py -m gpu -q -n 4 --timeout=600 2>&1 | tail -8 )
echo "== bench phase times"; ( timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --phase-times 2>&1 | tail -1 ) | tee gpurun_out/bench11.json | cut -c1-1700
echo "== align mode"; ( timeout 400 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 ) | tee gpurun_out/bench11_align.json | cut -c1-400
