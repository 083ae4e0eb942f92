#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== kernel + golden tests"; ( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -m gpu -q -n 4 --timeout=600 2>&1 | tail -15 ) > gpurun_out/gpu_tests10.log; tail -15 gpurun_out/gpu_tests10.log
echo "== bench phase times"; ( timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --phase-times 2>&1 | tail -1 ) | tee gpurun_out/bench10.json | cut -c1-3300
echo "== bench, first-generation-like double-buffered GEMM off (register-staged)"; ( SWX_FLAGS=$((4|16|64|512)) timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
echo "== align mode"; ( timeout 400 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 ) | tee gpurun_out/bench10_align.json | cut -c1-400
echo "== dec gemm cold vs hot weights"
DEC_HOT=0 scripts/rocprof_kernels.sh dec_cold python $R/scripts/dec_ablate.py; grep gemm_dec gpurun_out/dec_cold_kernels.csv
DEC_HOT=1 scripts/rocprof_kernels.sh dec_hot python $R/scripts/dec_ablate.py; grep gemm_dec gpurun_out/dec_hot_kernels.csv
