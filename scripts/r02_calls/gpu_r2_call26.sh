#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "ROC_ACTIVE_WAIT_TIMEOUT=1000000" "ROC_ACTIVE_WAIT_TIMEOUT=20000"; do
  echo "=== $cfg"; ( env $cfg timeout 300 python scripts/micro/align_trace.py 2>&1 | tail -8 | cut -c1-420 )
  echo "== align $cfg"; ( env $cfg timeout 600 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
  echo "== bench $cfg"; ( env $cfg timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
done 2>&1 | tee gpurun_out/active_wait_ab.txt
