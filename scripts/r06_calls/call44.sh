#!/bin/bash
# round 6 call 44: the fused query projection of the decode cross-attention stages only the window's nq rows of the residual tile (5 beams = 13
# of 40 LDS-DMA instructions; flag 512 = SWX_FLAG_XQ_FULL_TILE = all 16 rows): bit-identity (model / golden / batch invariance / bench
# windows), A/B on the headline pass, at 120 windows, in the sequential and span modes
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_batch_invariance.py tests/test_gpu_f16_bench_windows.py -m gpu -q -x 2>&1 | tail -3 ) > gpurun_out/r06_c44_tests.log; cat gpurun_out/r06_c44_tests.log
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 512 > gpurun_out/r06_c44_bench_xq_rows_ab.json 2> gpurun_out/r06_c44.err )
( timeout 600 python bench.py --minutes 60 --batch 120 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 512 > gpurun_out/r06_c44_bench_b120_xq_rows_ab.json 2>> gpurun_out/r06_c44.err )
( timeout 900 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 512 > gpurun_out/r06_c44_bench_seq_xq_rows_ab.json 2>> gpurun_out/r06_c44.err )
( timeout 600 python bench.py --spans 20 --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 512 > gpurun_out/r06_c44_bench_spans_xq_rows_ab.json 2>> gpurun_out/r06_c44.err )
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c44_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c44_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 gpurun_out/r06_c44.err
