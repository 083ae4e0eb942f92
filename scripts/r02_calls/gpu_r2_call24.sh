#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "HSA_ENABLE_INTERRUPT=0" "HSA_ENABLE_INTERRUPT=1"; do
  echo "=== $cfg"; ( env $cfg timeout 300 python scripts/micro/align_trace.py 2>&1 | tail -8 | cut -c1-400 )
  echo "== align $cfg"; ( env $cfg timeout 600 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
  echo "== bench $cfg"; ( env $cfg timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
  echo "== sequential $cfg"; ( env $cfg timeout 400 python bench.py --sequential --minutes 3 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
done 2>&1 | tee gpurun_out/hsa_interrupt_ab.txt
