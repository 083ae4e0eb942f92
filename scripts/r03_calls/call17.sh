#!/bin/bash
# round 3, call 17: the measurement set on the final code (PMC passes, rocprofv3 summary, default bench line with roofline /
# strict f32 / CPU baseline, smoke, the whole GPU suite) + the lines of the other BASELINE configs
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c17
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
echo "host kernel $(uname -r)" | tee $O/box.txt
bash scripts/gpu_final.sh r03final2 2>&1 | tail -60
timeout 500 python bench.py --mode align --steps 3 --warmup 1 --no-f32 > $O/bench_align.json 2>> $O/bench.err
timeout 400 python bench.py --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 20 --warmup 3 --no-f32 > $O/bench_base_en_1win.json 2>> $O/bench.err
timeout 500 python bench.py --minutes 60 --batch 120 --steps 2 --warmup 1 --no-f32 --no-cpu-baseline > $O/bench_60min_b120.json 2>> $O/bench.err
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], j['ms_per_step'], j['value'], (j.get('roofline') or {}).get('kernel'), (j.get('roofline') or {}).get('frac'), (j.get('cpu_baseline') or {}).get('value'))"; done
