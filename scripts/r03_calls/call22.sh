#!/bin/bash
# round 3, call 22: rocprofv3 --kernel-trace --stats summary of the bench command on the final tree (per-kernel average durations)
cd "$(dirname "$0")/../.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
echo "host kernel $(uname -r)"
scripts/rocprof_kernels.sh bench_final3 python $PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
head -14 gpurun_out/bench_final3_kernels.csv | cut -c1-140
tail -2 gpurun_out/bench_final3_cmd.log | cut -c1-300
