#!/bin/bash
mkdir -p gpurun_out
for B in 160 320 480; do
  echo "SWX_PG_BLOCKS=$B" >> gpurun_out/tune.log
  SWX_PG_BLOCKS=$B timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_time_ms'])" >> gpurun_out/tune.log
done
cat gpurun_out/tune.log
