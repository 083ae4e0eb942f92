#!/bin/bash
# GPU call 20: where the strict-f32 pass goes (kernel trace of one pass of the headline workload in dtype f32, 24 decode steps)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
scripts/rocprof_kernels.sh r04_c20_f32 python $R/bench.py --dtype f32 --tokens 24 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-f32
head -28 gpurun_out/r04_c20_f32_kernels.csv | cut -c1-170
head -3 gpurun_out/r04_c20_f32_gaps.csv
