#!/bin/bash
# round 6 call 22: strict f32 pass, lane exchanges on the VALU (product) vs ds_bpermute (-DSWX_LANE_XOR_BPERMUTE build of the same tree), alternating
mkdir -p gpurun_out
export TMPDIR=/tmp
cp stable_ts_amd/libswx.so /tmp/libswx_new.so
for lib in bpermute new bpermute new; do
    if [ $lib = bpermute ]; then cp scripts/exp/libswx_bpermute.so stable_ts_amd/libswx.so; else cp /tmp/libswx_new.so stable_ts_amd/libswx.so; fi
    ( timeout 900 python bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) >> gpurun_out/r06_c22_f32_${lib}.txt 2>> gpurun_out/r06_c22.err
done
cp /tmp/libswx_new.so stable_ts_amd/libswx.so
python - <<'PY'
import json
for l in ("bpermute", "new"):
    rows = [json.loads(x) for x in open(f"gpurun_out/r06_c22_f32_{l}.txt") if x.startswith("{")]
    print("strict f32", l, [r["ms_per_step"] for r in rows], [r["config"].get("words") for r in rows])
PY
