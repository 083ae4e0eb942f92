#!/bin/bash
# round 3, call 18: call 17 landed on a slow box (host kernel 6.18.50: profiles/r03_box_variance.txt); the default bench line and
# the 60-min / 120-window line of the final code once more
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c18
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
echo "host kernel $(uname -r)" | tee $O/box.txt
timeout 900 python bench.py > $O/bench_default.json 2>> $O/bench.err
timeout 500 python bench.py --minutes 60 --batch 120 --steps 2 --warmup 1 --no-f32 --no-cpu-baseline > $O/bench_60min_b120.json 2>> $O/bench.err
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], j['ms_per_step'], j['value'], (j.get('roofline') or {}).get('kernel'), (j.get('roofline') or {}).get('frac'), (j.get('cpu_baseline') or {}).get('value'), (j.get('strict_f32') or {}).get('value'))"; done
