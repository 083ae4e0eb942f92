#!/bin/bash
# round 6 call 43: single-wave dec GEMM workgroups stage only the tile rows that exist (5 rows = 13 of 40 DMA instructions; flag 256 =
# SWX_FLAG_DEC_W1_FULL_TILE = all 16 rows): bit-identity (vs the four-wave workgroups, per epilogue; model level), A/B in the sequential
# mode, on base.en's single window and in the span mode
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_batch_invariance.py -m gpu -q -x 2>&1 | tail -3 ) > gpurun_out/r06_c43_tests.log; cat gpurun_out/r06_c43_tests.log
( timeout 900 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 256 > gpurun_out/r06_c43_bench_seq_w1_rows_ab.json 2> gpurun_out/r06_c43.err )
( timeout 600 python bench.py --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 256 > gpurun_out/r06_c43_bench_base_en_w1_rows_ab.json 2>> gpurun_out/r06_c43.err )
( timeout 600 python bench.py --spans 20 --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 256 > gpurun_out/r06_c43_bench_spans_w1_rows_ab.json 2>> gpurun_out/r06_c43.err )
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c43_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c43_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 gpurun_out/r06_c43.err
