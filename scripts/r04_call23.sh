#!/bin/bash
# GPU call 23 (what is left of the round's GPU budget): the final tree -- gemm_f16_big8 as the 256 x 256 kernel of the dispatch,
# sample-exact temperature > 0 decoding -- (1) the complete default bench line (roofline, CPU baseline, strict-f32 leg),
# (2) the GPU tests the earlier subset of this round did not cover on this tree (large-v3 depth / batch invariance / kernels),
# (3) one GPU's share of the 8-GPU job (60 min at 120 windows per batch), where the encoder GEMMs weigh most.
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { echo "== $1 ($(date +%T))"; }
step "default bench line"; ( timeout 240 python bench.py 2>&1 | tail -1 ) | tee gpurun_out/r04_c23_bench_final.json | cut -c1-420
step "gpu tests"; ( timeout 330 python -m pytest tests/test_gpu_batch_invariance.py tests/test_gpu_kernels.py tests/test_gpu_largev3.py tests/test_gpu_f16_depth.py tests/test_gpu_dist.py -m gpu -q -n 4 --timeout=320 --tb=short -rf 2>&1 | tail -25 ) | tee gpurun_out/r04_c23_gpu_tests.log | tail -8
step "60 min at 120 windows per batch"; ( timeout 200 python bench.py --minutes 60 --batch 120 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 2>&1 | tail -1 ) | tee gpurun_out/r04_c23_bench_60min_b120.json | cut -c1-300
step "done"
