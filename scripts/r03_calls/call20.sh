#!/bin/bash
# round 3, call 20: the headline command on the final tree (host-side word-probability change included), with stage times
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c20
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
echo "host kernel $(uname -r)" | tee $O/box.txt
timeout 170 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline --phase-times > $O/bench_default.json 2>> $O/bench.err
python -c "
import json; j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], json.dumps(j.get('phase_ms')))"
