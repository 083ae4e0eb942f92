#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "SWX_DTW_ABL=0" "SWX_DTW_ABL=1" "SWX_DTW_ABL=2" "SWX_DTW_ABL=3"; do
  echo "== dtw timing $cfg"; ( env $cfg timeout 100 python scripts/kernel_bench.py --only dtw 2>&1 | grep "W=" )
done | tee gpurun_out/kb_dtw_gen3_abl.txt
