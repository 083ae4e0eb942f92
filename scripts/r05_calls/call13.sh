#!/bin/bash
# Round 5, GPU call 13: the in-launch slab reduction (second form) at 5 rows: sequential mode A/B inside one process
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 400 python bench.py --sequential --no-f32 --no-cpu-baseline --no-roofline --steps 1 --warmup 1 --ab-flags 2097152 2>gpurun_out/r05_c13.err | tail -1 ) > gpurun_out/r05_c13_bench_sequential_ticket_ab.json
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r05_c13_bench_sequential_ticket_ab.json'))
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'ab')})
except Exception as e:
    print("no bench line:", e); print(open('gpurun_out/r05_c13.err').read()[-600:])
PY
