#!/bin/bash
# GPU call 6: kernel-level profile of the headline pass (new dispatch) + timeline of two overlapped half-batch decode loops
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
scripts/rocprof_kernels.sh r04_c6_bench python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
head -45 gpurun_out/r04_c6_bench_kernels.csv | cut -c1-160
scripts/rocprof_kernels.sh r04_c6_lanes2 python $R/scripts/ab_decode_lanes.py --lanes 2 --reps 1
head -14 gpurun_out/r04_c6_lanes2_kernels.csv | cut -c1-160
head -3 gpurun_out/r04_c6_lanes2_gaps.csv
