#!/bin/bash
# round 6 call 39: f16 flash attention with 16 queries per wave for launches of <= 256 workgroups at 32 (the encoder of one window: 240 -> 480
# workgroups; flag 64 = SWX_FLAG_FLASH_NO_QB1): bit-identity, A/B in align(), the sequential mode and on base.en's single window
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3 ) > gpurun_out/r06_c39_tests.log; cat gpurun_out/r06_c39_tests.log
( timeout 600 python bench.py --mode align --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 64 > gpurun_out/r06_c39_bench_align_flash_qb1_ab.json 2> gpurun_out/r06_c39.err )
( timeout 900 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 64 > gpurun_out/r06_c39_bench_seq_flash_qb1_ab.json 2>> gpurun_out/r06_c39.err )
( timeout 600 python bench.py --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 64 > gpurun_out/r06_c39_bench_base_en_flash_qb1_ab.json 2>> gpurun_out/r06_c39.err )
( timeout 600 python scripts/kernel_bench.py --only flash_small --flags 0 2>&1 | tail -8 ) > gpurun_out/r06_c39_kb_flash_small.txt
( timeout 600 python scripts/kernel_bench.py --only flash_small --flags 64 2>&1 | tail -8 ) >> gpurun_out/r06_c39_kb_flash_small.txt
cat gpurun_out/r06_c39_kb_flash_small.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c39_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c39_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 gpurun_out/r06_c39.err
