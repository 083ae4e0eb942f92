#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
scripts/rocprof_kernels.sh align python $R/bench.py --mode align --minutes 10 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
head -45 gpurun_out/align_kernels.csv | cut -c1-200
tail -2 gpurun_out/align_cmd.log | cut -c1-400
