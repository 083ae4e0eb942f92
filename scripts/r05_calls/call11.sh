#!/bin/bash
# Round 5, GPU call 11: kernel summary of the pass WITH the in-launch slab reduction (SWX_FLAG_TICKET), to show where its 19 ms go
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash scripts/rocprof_kernels.sh r05_ticket_on python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --debug-flags 2097152
tail -1 gpurun_out/r05_ticket_on_cmd.log | cut -c1-200; head -14 gpurun_out/r05_ticket_on_kernels.csv
