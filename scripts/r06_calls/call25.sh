#!/bin/bash
# round 6 call 25: strict f32 pass after the SPLIT f32 flash kernel got its LDS exchanges back (its ISA equals the -DSWX_LANE_XOR_BPERMUTE build's again):
# product vs that build, alternating; attention + model tests; one headline line
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3 ) > gpurun_out/r06_c25_tests.log; cat gpurun_out/r06_c25_tests.log
cp stable_ts_amd/libswx.so /tmp/libswx_new.so
for lib in bpermute new bpermute new; do
    if [ $lib = bpermute ]; then cp scripts/exp/libswx_bpermute.so stable_ts_amd/libswx.so; else cp /tmp/libswx_new.so stable_ts_amd/libswx.so; fi
    ( timeout 900 python bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) >> gpurun_out/r06_c25_f32_${lib}.txt 2>> gpurun_out/r06_c25.err
done
cp /tmp/libswx_new.so stable_ts_amd/libswx.so
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline > gpurun_out/r06_c25_bench.json 2>> gpurun_out/r06_c25.err )
python - <<'PY'
import json
for l in ("bpermute", "new"):
    rows = [json.loads(x) for x in open(f"gpurun_out/r06_c25_f32_{l}.txt") if x.startswith("{")]
    print("strict f32", l, [r["ms_per_step"] for r in rows], [r["config"].get("words") for r in rows])
d = json.load(open("gpurun_out/r06_c25_bench.json")); print("headline", d["value"], d["ms_per_step"], d["config"]["words"])
PY
