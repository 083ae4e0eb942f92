#!/bin/bash
# round 3, call 21: the align() line on the final tree
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c21
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
echo "host kernel $(uname -r)" | tee $O/box.txt
timeout 100 python bench.py --mode align --steps 3 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_align.json 2>> $O/bench.err
python -c "
import json; j=json.loads(open('$O/bench_align.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['config'].get('encoder_calls_per_pass'))"
