#!/bin/bash
# round 6 call 45: the vocabulary projection of < 128 rows stages only the A-tile chunks that hold a row (gemm_f16_glds_128_few; flag 1024 =
# SWX_FLAG_GEMM_NO_FEW): bit-identity, A/B in the sequential mode, on base.en's single window, on the headline pass
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_golden.py -m gpu -q -x 2>&1 | tail -3 ) > gpurun_out/r06_c45_tests.log; cat gpurun_out/r06_c45_tests.log
( timeout 900 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 1024 > gpurun_out/r06_c45_bench_seq_logits_few_ab.json 2> gpurun_out/r06_c45.err )
( timeout 600 python bench.py --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 1024 > gpurun_out/r06_c45_bench_base_en_logits_few_ab.json 2>> gpurun_out/r06_c45.err )
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 1024 > gpurun_out/r06_c45_bench_logits_few_ab.json 2>> gpurun_out/r06_c45.err )
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c45_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c45_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 gpurun_out/r06_c45.err
