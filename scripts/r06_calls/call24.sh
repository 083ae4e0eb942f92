#!/bin/bash
# round 6 call 24: the HBM-streaming attention kernels' row exchanges back on ds_bpermute (product) vs on the VALU (-DSWX_XATTN_VALU build = the
# tree of calls 13-23): headline pass and strict f32 pass, alternating on one box; attention kernel tests
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention or lane_xor" 2>&1 | tail -4 ) > gpurun_out/r06_c24_tests.log; cat gpurun_out/r06_c24_tests.log
cp stable_ts_amd/libswx.so /tmp/libswx_new.so
for lib in xattn_valu new xattn_valu new; do
    if [ $lib = xattn_valu ]; then cp scripts/exp/libswx_xattn_valu.so stable_ts_amd/libswx.so; else cp /tmp/libswx_new.so stable_ts_amd/libswx.so; fi
    ( timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline ) >> gpurun_out/r06_c24_bench_${lib}.txt 2>> gpurun_out/r06_c24.err
    ( timeout 900 python bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) >> gpurun_out/r06_c24_f32_${lib}.txt 2>> gpurun_out/r06_c24.err
    ( timeout 600 python scripts/kernel_bench.py --only cross --iters 200 ) >> gpurun_out/r06_c24_kb_cross_${lib}.txt 2>> gpurun_out/r06_c24.err
done
cp /tmp/libswx_new.so stable_ts_amd/libswx.so
python - <<'PY'
import json
for t in ("bench", "f32"):
    for l in ("xattn_valu", "new"):
        rows = [json.loads(x) for x in open(f"gpurun_out/r06_c24_{t}_{l}.txt") if x.startswith("{")]
        print(t, l, [r["ms_per_step"] for r in rows], [r["config"].get("words") for r in rows])
for l in ("xattn_valu", "new"):
    print("kb_cross", l, [x.strip() for x in open(f"gpurun_out/r06_c24_kb_cross_{l}.txt") if "us" in x])
PY
