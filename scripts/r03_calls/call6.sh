#!/bin/bash
# round 3, call 6: prefetch chain v2 (all six projections) A/B; flash attention at 32 / 48 / 64 queries per wave
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c6
mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
stamp() { echo "=== $1 (t+$(( $(date +%s) - t0 ))s)"; }
stamp "kernel tests (flash variants, prefetch identity, graph)"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -k "attention_kernels or prefetch or graph_replay" -p no:cacheprovider > $O/pytest.log 2>&1
tail -4 $O/pytest.log
stamp "flash attention: queries per wave"
timeout 300 python scripts/kernel_bench.py --only flash --iters 100 > $O/kb_flash.txt 2>&1; cat $O/kb_flash.txt | tail -8
stamp "A/B default bench (prefetch on / off / on / off)"
for F in 0 32768 0 32768; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline --debug-flags $F > $O/bench_default_f${F}_$(date +%s).json 2>> $O/bench.err
done
for f in $O/bench_default_f*.json; do python -c "import json,sys; j=json.load(open('$f')); print(j.get('debug_flags',0), j['ms_per_step'], j['value'])"; done
stamp "done"
