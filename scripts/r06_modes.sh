#!/bin/bash
# usage: gpurun --timeout 2400 -- 'bash scripts/r06_modes.sh [tag]'
# one driver-reproducible bench line per BASELINE config / mode on the final code (each a fresh process, same box) -- round 6: every line
# WITH its `roofline` (instrumented pass, HIP events inside libswx) and `cpu_baseline` (oracle port on the box's host cores) objects
# (VERDICT r5 missing 5: by the tier's rule a config without them is unmeasured); only the strict-f32 leg stays with the headline line.
tag=${1:-final}
mkdir -p gpurun_out
B="--no-f32 --cpu-budget 150"
run() { name=$1; shift; echo "== $name: bench.py $* ($(date +%T))"; ( timeout 600 python bench.py "$@" 2>gpurun_out/r06_${tag}_bench_$name.err | tail -1 ) | tee gpurun_out/r06_${tag}_bench_$name.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d.get('roofline',{}); b=d.get('cpu_baseline',{}); print(d['value'], 'x', d['ms_per_step'], 'ms', {k: c.get(k) for k in ('windows_per_gpu','words','encoder_calls_per_pass','text_tokens_per_window')}, 'roofline', r.get('kernel'), r.get('frac'), 'cpu', b.get('value'), b.get('cores'))"; }
run base_en_1win --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 10 --warmup 3 $B
run align --mode align --steps 2 --warmup 1 $B
run sequential --sequential --steps 1 --warmup 1 $B
run spans20 --spans 20 --steps 2 --warmup 1 $B
run host_audio --host-audio --steps 3 --warmup 1 $B
run sharded_w1 --mode sharded --steps 3 --warmup 1 $B
run 60min_b120 --minutes 60 --batch 120 --steps 1 --warmup 1 $B
