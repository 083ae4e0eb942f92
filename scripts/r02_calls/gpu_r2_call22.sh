#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "TRACE_VERBOSE=1" "TRACE_VERBOSE=1 SWX_DTW_GEN2=1" "TRACE_VERBOSE=1 SWX_FLAGS=340"; do
  echo "=== $cfg"; ( env $cfg timeout 300 python scripts/micro/align_trace.py 2>&1 | tail -14 | cut -c1-200 )
done 2>&1 | tee gpurun_out/align_trace.txt
