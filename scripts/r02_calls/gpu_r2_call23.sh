#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "A=1" "GPU_MAX_HW_QUEUES=1" "HIP_FORCE_DEV_KERNARG=1"; do
  echo "=== $cfg"; ( env $cfg timeout 300 python scripts/micro/align_trace.py 2>&1 | tail -8 | cut -c1-900 )
done 2>&1 | tee gpurun_out/align_trace2.txt
