#!/bin/bash
# round 6 call 21: GELU epilogue at realistic activation magnitudes -- 256 x 256 kernel vs 128 x 128 kernel
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python scripts/kernel_bench.py --only gemm_gelu --iters 100 ) > gpurun_out/r06_c21_kb_gemm_gelu.txt 2> gpurun_out/r06_c21_kb.err
cat gpurun_out/r06_c21_kb_gemm_gelu.txt; tail -3 gpurun_out/r06_c21_kb.err
