#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== dtw tests"; ( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -n 4 --timeout=300 -k "dtw" 2>&1 | tail -4 )
echo "== dtw timing"; ( timeout 100 python scripts/kernel_bench.py --only dtw 2>&1 | grep "W=" ) | tee gpurun_out/kb_dtw_gen3b.txt
scripts/rocprof_kernels.sh align python $R/bench.py --mode align --minutes 10 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
head -32 gpurun_out/align_gaps.csv | cut -c1-160
scripts/rocprof_kernels.sh trans python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
head -24 gpurun_out/trans_gaps.csv | cut -c1-160
