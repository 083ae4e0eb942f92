#!/bin/bash
# round 6 call 42: the fragment-ordered copy of the cross-attention K / V^T written by the K | V projection's epilogue (GemmArgs::P) instead of
# by swx_xkv_pack (flag 128 = SWX_FLAG_XKV_PACK_SEPARATE): the same bytes (test), model / golden / batch-invariance tests, A/B on the headline
# pass, in align() and at 120 windows
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_batch_invariance.py -m gpu -q -x 2>&1 | tail -3 ) > gpurun_out/r06_c42_tests.log; cat gpurun_out/r06_c42_tests.log
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 128 > gpurun_out/r06_c42_bench_xkv_pack_epilogue_ab.json 2> gpurun_out/r06_c42.err )
( timeout 600 python bench.py --mode align --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 128 > gpurun_out/r06_c42_bench_align_xkv_pack_epilogue_ab.json 2>> gpurun_out/r06_c42.err )
( timeout 600 python bench.py --minutes 60 --batch 120 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 128 > gpurun_out/r06_c42_bench_b120_xkv_pack_epilogue_ab.json 2>> gpurun_out/r06_c42.err )
( timeout 900 bash scripts/rocprof_kernels.sh r06_c42_pass python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) > gpurun_out/r06_c42_rocprof.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c42_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c42_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
grep "glds_128\|xkv_pack" gpurun_out/r06_c42_pass_kernels.csv | cut -c1-170
tail -3 gpurun_out/r06_c42.err
