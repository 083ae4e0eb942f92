#!/bin/bash
# round 6 call 1: the x1 closure tests (strict f32 on the timed configuration, f32 batch invariance, per-window divergence of the
# f16 / f32 passes), the ticket test on persistent counters, and the default bench line on this tree
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_pass_divergence.py tests/test_gpu_f16_bench_windows.py tests/test_gpu_batch_invariance.py \
    "tests/test_gpu_kernels.py::test_dec_gemm_slab_reduction_inside_the_launch_is_bit_identical" -m gpu -q -x --durations=15 2>&1 | tail -60 ) > gpurun_out/r06_c1_tests.log
( timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r06_c1_bench.json 2> gpurun_out/r06_c1_bench.err )
tail -30 gpurun_out/r06_c1_tests.log
cat gpurun_out/r06_c1_bench.json
