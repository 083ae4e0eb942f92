#!/bin/bash
# GPU call 10: where the sequential mode's time goes (kernel trace of 2 minutes = 4 windows) + phase times
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-f32 --no-roofline"
scripts/rocprof_kernels.sh r04_c10_seq python $R/bench.py --sequential --minutes 2 --steps 1 --warmup 1 $B
head -40 gpurun_out/r04_c10_seq_kernels.csv | cut -c1-170
head -12 gpurun_out/r04_c10_seq_gaps.csv
echo "== phase times"; ( timeout 300 python bench.py --sequential --minutes 2 --steps 1 --warmup 1 --phase-times $B 2>&1 | tail -1 ) | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(json.dumps(d.get('phase_ms'), indent=1))"
