#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
scripts/rocprof_kernels.sh align python $R/bench.py --mode align --minutes 10 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
head -3 gpurun_out/align_gaps.csv
awk -F, 'NR>1 && $5>8' gpurun_out/align_slice.csv | head -60 | cut -c1-150
