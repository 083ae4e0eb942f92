#!/bin/bash
# round 6 call 28: (a) decode cross-attention with FOUR key blocks of a wave in flight for launches of at most one workgroup per CU
# (attn_decode_cross_xq4_f16 / cross4_f16; flag 2 = SWX_FLAG_XATTN_NO_DEEP puts the two-block kernels back): bit-identity tests, A/B in the
# sequential mode (one window per decode call: 20 workgroups) and in align() (scoring pass of one window: 100-160 workgroups);
# (b) the scoring pass's vocabulary projection + token probabilities per GROUP of windows (flag 536870912 = per window): tests, A/B on the
# headline pass and at batch 120
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_batch_invariance.py -m gpu -q -x 2>&1 | tail -5 ) > gpurun_out/r06_c28_tests.log; cat gpurun_out/r06_c28_tests.log
( timeout 900 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 2 > gpurun_out/r06_c28_bench_seq_xattn_deep_ab.json 2> gpurun_out/r06_c28.err )
( timeout 600 python bench.py --mode align --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 2 > gpurun_out/r06_c28_bench_align_xattn_deep_ab.json 2>> gpurun_out/r06_c28.err )
( timeout 600 python bench.py --mode align --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 536870912 > gpurun_out/r06_c28_bench_align_score_group_ab.json 2>> gpurun_out/r06_c28.err )
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 536870912 > gpurun_out/r06_c28_bench_score_group_ab.json 2>> gpurun_out/r06_c28.err )
( timeout 600 python bench.py --minutes 60 --batch 120 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 536870912 > gpurun_out/r06_c28_bench_b120_score_group_ab.json 2>> gpurun_out/r06_c28.err )
( timeout 600 python bench.py --spans 20 --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 2 > gpurun_out/r06_c28_bench_spans_xattn_deep_ab.json 2>> gpurun_out/r06_c28.err )
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c28_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c28_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -5 gpurun_out/r06_c28.err
