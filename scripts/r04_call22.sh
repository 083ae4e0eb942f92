#!/bin/bash
# GPU call 22 (the round's last ~10 GPU-minutes): first hardware run of
#   (1) gemm_f16_big8 (half-tile ring, staggered wave groups; force_kernel 13 / SWX_FLAG_BIG8 = 2097152): bit-identity with
#       gemm_f16_big / gemm_f16_tiled under background traffic, micro-benchmark by force_kernel, headline pass A/B by flag,
#       rocprofv3 kernel summary of the pass with the flag on;
#   (2) sample-exact temperature > 0 decoding (Engine.decode(torch_rng=True)) against the oracle sampling on the GPU generator;
#   (3) the parts of the GPU suite the two changes touch (tiled-GEMM checks, decode / sampling tests, goldens).
# Every step has its own timeout and the script goes on after a failure: one call, no second chance.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
step() { echo "== $1 ($(date +%T))"; }
step "big8 check"; timeout 200 python tests/hw_checks/gemm_big8_check.py --reps 20 > gpurun_out/r04_c22_big8_check.txt 2>&1; big8_rc=$?
tail -18 gpurun_out/r04_c22_big8_check.txt; echo "big8 check exit code $big8_rc" | tee -a gpurun_out/r04_c22_big8_check.txt
FL=0; [ $big8_rc -eq 0 ] && FL=2097152
step "kernel bench gemm_big"; ( timeout 120 python scripts/kernel_bench.py --only gemm_big 2>&1 | tail -18 ) | tee gpurun_out/r04_c22_kb_gemm_big.txt
step "sampling test"; ( SWX_INNER_TESTS=1 timeout 200 python -m pytest "tests/test_gpu_golden.py::test_inner_sampled_decoding_follows_torch_generator" -q -x -m gpu -p no:cacheprovider 2>&1 | tail -25 ) | tee gpurun_out/r04_c22_sampling_test.txt
step "bench A/B big8 in one process"; ( timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --ab-flags 2097152 2>&1 | tail -1 ) | tee gpurun_out/r04_c22_bench_ab_big8.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ab'))"
step "rocprof of the pass (debug flags $FL)"; scripts/rocprof_kernels.sh r04_c22_pass python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --debug-flags $FL
head -14 gpurun_out/r04_c22_pass_kernels.csv | cut -c1-150
step "gpu tests (subset)"; ( timeout 420 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_model.py -m gpu -q -n 3 --timeout=400 --tb=short -rf 2>&1 | tail -25 ) | tee gpurun_out/r04_c22_gpu_subset.log | tail -8
step "done"
