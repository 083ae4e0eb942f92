#!/bin/bash
# round 6 call 40: vocabulary projection of <= 128 rows on 224-column ring tiles (one round of 232 workgroups; flag 128 = SWX_FLAG_NO_RING224 =
# 406 tiles of 128 columns on the occupancy-overlapped kernel): bit-identity, A/B on the headline pass, in the sequential mode, base.en
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_batch_invariance.py -m gpu -q -x 2>&1 | tail -3 ) > gpurun_out/r06_c40_tests.log; cat gpurun_out/r06_c40_tests.log
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 128 > gpurun_out/r06_c40_bench_logits_ring224_ab.json 2> gpurun_out/r06_c40.err )
( timeout 900 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 128 > gpurun_out/r06_c40_bench_seq_logits_ring224_ab.json 2>> gpurun_out/r06_c40.err )
( timeout 600 python bench.py --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 128 > gpurun_out/r06_c40_bench_base_en_logits_ring224_ab.json 2>> gpurun_out/r06_c40.err )
( timeout 600 python bench.py --minutes 60 --batch 120 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 128 > gpurun_out/r06_c40_bench_b120_logits_ring224_ab.json 2>> gpurun_out/r06_c40.err )
( timeout 900 bash scripts/rocprof_kernels.sh r06_c40_pass python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) > gpurun_out/r06_c40_rocprof.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c40_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c40_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
grep "gemm_f16_ring\|glds_128" gpurun_out/r06_c40_pass_kernels.csv | cut -c1-170
tail -3 gpurun_out/r06_c40.err
