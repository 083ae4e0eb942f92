#!/bin/bash
# round 6 call 34: scoring-pass cross-attention of many windows (attn_decode_cross2_f16<true, 4>: 280 registers = ONE workgroup per CU = one wave
# per SIMD, 98 us per layer): (a) two groups of 16 rows per workgroup instead of four (flag 64: 232 registers, two workgroups per CU, the
# head's K / V^T streamed four times instead of twice); (b) four groups held to 256 registers by __launch_bounds__(256, 2) (flag 128: 234
# registers, no spill).  Both bit-identical by construction (same body); A/B on the headline pass and at 120 windows
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_batch_invariance.py -m gpu -q -x 2>&1 | tail -3 ) > gpurun_out/r06_c34_tests.log; cat gpurun_out/r06_c34_tests.log
for F in 64 128; do
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags $F > gpurun_out/r06_c34_bench_xattn_qg_f${F}_ab.json 2>> gpurun_out/r06_c34.err )
( timeout 600 python bench.py --minutes 60 --batch 120 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags $F > gpurun_out/r06_c34_bench_b120_xattn_qg_f${F}_ab.json 2>> gpurun_out/r06_c34.err )
done
( timeout 900 bash scripts/rocprof_kernels.sh r06_c34_f128 python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --debug-flags 128 ) > gpurun_out/r06_c34_rocprof.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c34_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c34_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
grep "attn_decode_cross2" gpurun_out/r06_c34_f128_kernels.csv | cut -c1-170
tail -3 gpurun_out/r06_c34.err
