#!/bin/bash
# round 3, call 8: repeat of call 7's A/B (that box ran 35 % slow and noisy): 8 vs 4 waves in the decode cross-attention
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c8
mkdir -p $O
export PYTHONUNBUFFERED=1
cp stable_ts_amd/libswx.so /tmp/libswx_base.so
for V in base xw8 base xw8 base xw8; do
  if [ $V = base ]; then cp /tmp/libswx_base.so stable_ts_amd/libswx.so; else cp scripts/exp/libswx_$V.so stable_ts_amd/libswx.so; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_${V}_$(date +%s).json 2>> $O/bench.err
done
cp /tmp/libswx_base.so stable_ts_amd/libswx.so
for f in $O/bench_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], j['ms_per_step'], j['value'])"; done
