#!/bin/bash
# round 3, call 4: L2 prefetch of the next projection's weights -- bit-identity, A/B (bench default / sequential / base.en),
# per-kernel durations with and without (rocprofv3, eager launches)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c4
mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
stamp() { echo "=== $1 (t+$(( $(date +%s) - t0 ))s)"; }
stamp "bit identity"
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_largev3.py -q -m gpu -k "prefetch or graph_replay or unsaturated" -p no:cacheprovider > $O/pytest.log 2>&1
tail -4 $O/pytest.log
stamp "A/B default bench (prefetch on / off / on / off)"
for F in 0 32768 0 32768; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline --debug-flags $F > $O/bench_default_f${F}_$(date +%s).json 2>> $O/bench.err
done
grep -h -o '"value": [0-9.]*, "unit": "x real time", "n_gpus": 1, "n_gpus_measured": 1, "steps": 6, "warmup": 2, "ms_per_step": [0-9.]*\|"debug_flags": [0-9]*' $O/bench_default_f*.json
stamp "A/B sequential + base.en"
for F in 0 32768; do
  timeout 300 python bench.py --sequential --steps 1 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline --debug-flags $F > $O/bench_sequential_f$F.json 2>> $O/bench.err
  head -c 230 $O/bench_sequential_f$F.json; echo
  timeout 300 python bench.py --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 20 --warmup 3 --no-f32 --no-cpu-baseline --no-roofline --debug-flags $F > $O/bench_base_en_f$F.json 2>> $O/bench.err
  head -c 230 $O/bench_base_en_f$F.json; echo
done
stamp "per-kernel durations, eager launches, prefetch on / off"
scripts/rocprof_kernels.sh r03_eager_pf_on python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --debug-flags 16384
scripts/rocprof_kernels.sh r03_eager_pf_off python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --debug-flags 49152
head -14 gpurun_out/r03_eager_pf_on_kernels.csv | cut -c1-150
head -14 gpurun_out/r03_eager_pf_off_kernels.csv | cut -c1-150
stamp "done"
