#!/bin/bash
# round 6 call 33: (a) the tall dec GEMM with EIGHT waves = two 64-column panels per workgroup (flag 32 = SWX_FLAG_TALL_W8): bit-identity
# (dec_tall_check: reference launch / tall / tall with eight waves, 16 repetitions per shape), A/B on the headline pass, at 120 windows and in
# the span mode; (b) model-level bit-identity of the few-workgroup kernels (test_decode_f16_few_workgroup_kernels_are_bit_identical)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python tests/hw_checks/dec_tall_check.py 2>&1 | tail -16 ) > gpurun_out/r06_c33_dec_tall_check.txt; cat gpurun_out/r06_c33_dec_tall_check.txt
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "few_workgroup or graph_replay" 2>&1 | tail -3 ) > gpurun_out/r06_c33_tests.log; cat gpurun_out/r06_c33_tests.log
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 32 > gpurun_out/r06_c33_bench_tall_w8_ab.json 2> gpurun_out/r06_c33.err )
( timeout 600 python bench.py --minutes 60 --batch 120 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 32 > gpurun_out/r06_c33_bench_b120_tall_w8_ab.json 2>> gpurun_out/r06_c33.err )
( timeout 600 python bench.py --spans 20 --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 32 > gpurun_out/r06_c33_bench_spans_tall_w8_ab.json 2>> gpurun_out/r06_c33.err )
( timeout 900 bash scripts/rocprof_kernels.sh r06_c33_w8 python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --debug-flags 32 ) > gpurun_out/r06_c33_rocprof.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c33_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c33_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
grep dectall gpurun_out/r06_c33_w8_kernels.csv | cut -c1-170
tail -3 gpurun_out/r06_c33.err
