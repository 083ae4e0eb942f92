#!/bin/bash
# usage: gpurun --timeout 1500 -- 'bash scripts/r05_modes.sh [tag]'
# one driver-reproducible bench line per BASELINE config / mode on the final code (each a fresh process, same box)
tag=${1:-final}
mkdir -p gpurun_out
B="--no-cpu-baseline --no-f32 --no-roofline"
run() { name=$1; shift; echo "== $name: bench.py $* ($(date +%T))"; ( timeout 400 python bench.py "$@" 2>&1 | tail -1 ) | tee gpurun_out/r05_${tag}_bench_$name.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], 'x', d['ms_per_step'], 'ms', {k: c.get(k) for k in ('windows_per_gpu','words','encoder_calls_per_pass','text_tokens_per_window','decode_loop')})"; }
run base_en_1win --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 10 --warmup 3 $B
run align --mode align --steps 2 --warmup 1 $B
run sequential --sequential --steps 1 --warmup 1 $B
run spans20 --spans 20 --steps 2 --warmup 1 $B
run host_audio --host-audio --steps 3 --warmup 1 $B
run sharded_w1 --mode sharded --steps 3 --warmup 1 $B
run 60min_b120 --minutes 60 --batch 120 --steps 1 --warmup 1 $B
