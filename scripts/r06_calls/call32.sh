#!/bin/bash
# round 6 call 32: decode cross-attention of few windows split over four single-wave workgroups + ticket merge (attn_decode_cross_split_f16,
# query projection as its own launch; flag 2 = SWX_FLAG_XATTN_NO_SPLIT = the one-workgroup kernel with the fused projection): bit-identity,
# A/B in the sequential mode and on base.en's single window, kernel table
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_batch_invariance.py -m gpu -q -x 2>&1 | tail -5 ) > gpurun_out/r06_c32_tests.log; cat gpurun_out/r06_c32_tests.log
( timeout 900 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 2 > gpurun_out/r06_c32_bench_seq_xattn_split_ab.json 2> gpurun_out/r06_c32.err )
( timeout 600 python bench.py --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 2 > gpurun_out/r06_c32_bench_base_en_xattn_split_ab.json 2>> gpurun_out/r06_c32.err )
( timeout 900 bash scripts/rocprof_kernels.sh r06_c32_seq python $R/bench.py --sequential --minutes 2 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) > gpurun_out/r06_c32_rocprof.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c32_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c32_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
head -12 gpurun_out/r06_c32_seq_kernels.csv | cut -c1-170
tail -3 gpurun_out/r06_c32.err
