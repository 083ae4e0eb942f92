#!/bin/bash
# GPU call 9: bench lines of the other modes on the round-4 code
mkdir -p gpurun_out
B="--no-cpu-baseline --no-f32 --no-roofline"
echo "== sequential (fair seek)"; ( timeout 400 python bench.py --sequential --steps 1 --warmup 1 $B 2>&1 | tail -1 ) | tee gpurun_out/r04_c9_bench_sequential.json | cut -c1-900
echo "== spans 20 (fair seek)"; ( timeout 300 python bench.py --spans 20 --steps 2 --warmup 1 $B 2>&1 | tail -1 ) | tee gpurun_out/r04_c9_bench_spans20.json | cut -c1-700
echo "== spans 10 (fair seek)"; ( timeout 300 python bench.py --spans 10 --steps 2 --warmup 1 $B 2>&1 | tail -1 ) | tee gpurun_out/r04_c9_bench_spans10.json | cut -c1-500
echo "== host audio"; ( timeout 300 python bench.py --host-audio --steps 3 --warmup 1 $B 2>&1 | tail -1 ) | tee gpurun_out/r04_c9_bench_host_audio.json | cut -c1-500
echo "== align"; ( timeout 300 python bench.py --mode align --steps 1 --warmup 1 $B 2>&1 | tail -1 ) | tee gpurun_out/r04_c9_bench_align.json | cut -c1-500
