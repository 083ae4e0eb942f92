#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== glds hw check"; timeout 300 python tests/hw_checks/gemm_glds_check.py 2>&1 | tail -12
for k in 4 5 6 7; do echo "== gemm kernel $k"; timeout 200 python scripts/kernel_bench.py --only gemm --gemm-kernel $k 2>&1 | grep "M=" ; done | tee gpurun_out/kb_gemm_gen2.txt
echo "== model tests (select kernel identity, small dec pass)"; ( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -n 4 --timeout=600 -k "select or small_pass or dec_step or strict" 2>&1 | tail -8 )
echo "== bench phase times"; ( timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --phase-times 2>&1 | tail -1 ) | tee gpurun_out/bench9.json | cut -c1-3000
echo "== bench, select in memory"; ( SWX_FLAGS=$((4|16|64|512|256|8192)) timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
echo "== dec gemm cold vs hot weights"
DEC_HOT=0 scripts/rocprof_kernels.sh dec_cold python scripts/dec_ablate.py; grep gemm_dec gpurun_out/dec_cold_kernels.csv
DEC_HOT=1 scripts/rocprof_kernels.sh dec_hot python scripts/dec_ablate.py; grep gemm_dec gpurun_out/dec_hot_kernels.csv
