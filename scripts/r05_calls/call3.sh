#!/bin/bash
# Round 5, GPU call 3: first hardware run of the strict-f32 kernels of this round (attn_flash_f32<VT, SPLIT>, gemm_f32_rows64<NJ>):
# (1) their kernel tests, (2) the whole GPU suite (every strict-f32 golden / oracle comparison runs on them now), (3) the default
# bench line incl. the strict-f32 leg, (4) rocprofv3 kernel summary of the f32 pass.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
step() { echo "== $1 ($(date +%T))"; }
step "kernel tests"
( timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=280 --tb=short -k "attention or gemm_f32" 2>&1 | tail -15 ) | tee gpurun_out/r05_c3_kernel_tests.log | cut -c1-300 | tail -12
step "gpu suite"
( timeout 900 python -m pytest tests -m gpu -q -n 4 --timeout=600 --tb=short -rf 2>&1 | tail -40 ) | tee gpurun_out/r05_c3_gpu_suite.log | cut -c1-400 | tail -30
step "bench"
( timeout 300 python bench.py 2>gpurun_out/r05_c3_bench.err | tail -1 ) | tee gpurun_out/r05_c3_bench.json | cut -c1-1500
step "f32 pass kernels"
bash scripts/rocprof_kernels.sh r05_c3_f32pass python $R/bench.py --dtype f32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
tail -1 gpurun_out/r05_c3_f32pass_cmd.log | cut -c1-300; head -24 gpurun_out/r05_c3_f32pass_kernels.csv
step done
