#!/bin/bash
# round 3, call 7: decode cross-attention with 8 waves per (window, head) (variant library) vs 4; prefetch in the small pass (align)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c7
mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
stamp() { echo "=== $1 (t+$(( $(date +%s) - t0 ))s)"; }
stamp "tests: cross-attention kernels, packed-vs-row identity, small pass"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -k "attention or packed_cross or small_pass or prefetch" -p no:cacheprovider > $O/pytest.log 2>&1
tail -3 $O/pytest.log
cp stable_ts_amd/libswx.so /tmp/libswx_base.so
for V in base xw8 base xw8; do
  if [ $V = base ]; then cp /tmp/libswx_base.so stable_ts_amd/libswx.so; else cp scripts/exp/libswx_$V.so stable_ts_amd/libswx.so; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_${V}_$(date +%s).json 2>> $O/bench.err
done
for f in $O/bench_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], j['ms_per_step'], j['value'])"; done
stamp "xw8: tests on the variant library"
cp scripts/exp/libswx_xw8.so stable_ts_amd/libswx.so
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -k "attention or packed_cross" -p no:cacheprovider > $O/pytest_xw8.log 2>&1
tail -3 $O/pytest_xw8.log
stamp "xw8 at batch 120 (60 min)"
timeout 400 python bench.py --minutes 60 --batch 120 --steps 2 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_60min_b120_xw8.json 2>> $O/bench.err
cp /tmp/libswx_base.so stable_ts_amd/libswx.so
timeout 400 python bench.py --minutes 60 --batch 120 --steps 2 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_60min_b120_base.json 2>> $O/bench.err
for f in $O/bench_60min*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], j['ms_per_step'], j['value'])"; done
stamp "align with the prefetch chain in the small pass"
timeout 400 python bench.py --mode align --steps 3 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_align.json 2>> $O/bench.err
timeout 400 python bench.py --mode align --steps 3 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline --debug-flags 32768 > $O/bench_align_nopf.json 2>> $O/bench.err
for f in $O/bench_align*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], j['ms_per_step'], j['value'])"; done
stamp "done"
