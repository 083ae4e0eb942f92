#!/bin/bash
# round 3, call 1: the new fp16-vs-oracle parity tests (full depth), the bench lines of every BASELINE config, batch scaling,
# nt A/B.  Everything lands in gpurun_out/r03_c1/.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c1
mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
stamp() { echo "=== $1 (t+$(( $(date +%s) - t0 ))s)"; }

stamp "parity tests: fp16 vs oracle"
timeout 900 python -m pytest tests/test_gpu_f16_depth.py tests/test_gpu_largev3.py tests/test_gpu_golden.py -q -m gpu \
    -k "f16 or sharp" -p no:cacheprovider > $O/pytest_f16.log 2>&1
tail -15 $O/pytest_f16.log
cp gpurun_out/f16_depth_report.json gpurun_out/f16_report.json gpurun_out/align_sharp_*.json $O/ 2>/dev/null

stamp "bench: default (configs[2]) with regroup in the timed pass"
timeout 600 python bench.py --steps 5 --warmup 2 --phase-times > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json

stamp "bench: configs[1] base.en, one 30-s window, greedy"
timeout 300 python bench.py --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 20 --warmup 3 > $O/bench_base_en_1win.json 2> $O/bench_base_en_1win.err
head -c 700 $O/bench_base_en_1win.json; echo

stamp "bench: configs[3] align()"
timeout 400 python bench.py --mode align --steps 3 --warmup 1 --no-f32 > $O/bench_align.json 2> $O/bench_align.err
head -c 700 $O/bench_align.json; echo

stamp "bench: nt weight loads (variant library)"
cp stable_ts_amd/libswx.so /tmp/libswx_base.so
cp scripts/exp/libswx_nt.so stable_ts_amd/libswx.so
timeout 300 python bench.py --steps 5 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_nt.json 2> $O/bench_nt.err
head -c 300 $O/bench_nt.json; echo
cp /tmp/libswx_base.so stable_ts_amd/libswx.so
timeout 300 python bench.py --steps 5 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_base_again.json 2> $O/bench_base_again.err
head -c 300 $O/bench_base_again.json; echo

stamp "bench: spans 20 / sequential"
timeout 300 python bench.py --spans 20 --steps 2 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_spans20.json 2> $O/bench_spans20.err
head -c 300 $O/bench_spans20.json; echo
timeout 300 python bench.py --sequential --steps 1 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_sequential.json 2> $O/bench_sequential.err
head -c 300 $O/bench_sequential.json; echo

stamp "batch scaling: 60 min per GPU (config 5's share), batch 40 / 60 / 120"
for B in 40 60 120; do
  timeout 400 python bench.py --minutes 60 --batch $B --steps 2 --warmup 1 --no-f32 --no-cpu-baseline --phase-times > $O/bench_60min_b$B.json 2> $O/bench_60min_b$B.err
  head -c 300 $O/bench_60min_b$B.json; echo
done
stamp "sharded mode at world 1"
timeout 300 python bench.py --mode sharded --steps 2 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_sharded_w1.json 2> $O/bench_sharded_w1.err
head -c 300 $O/bench_sharded_w1.json; echo
stamp "done"
