#!/bin/bash
# round 6 call 23: which kernel of the strict f32 pass is slower with the VALU lane exchanges? rocprofv3 kernel summary of one f32 pass per build
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp stable_ts_amd/libswx.so /tmp/libswx_new.so
for lib in bpermute new; do
    if [ $lib = bpermute ]; then cp scripts/exp/libswx_bpermute.so stable_ts_amd/libswx.so; else cp /tmp/libswx_new.so stable_ts_amd/libswx.so; fi
    bash scripts/rocprof_kernels.sh r06_c23_f32_$lib python $R/bench.py --dtype f32 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline
done
cp /tmp/libswx_new.so stable_ts_amd/libswx.so
python - <<'PY'
import csv
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[(r["name"][:70], r["grid_x"], r["grid_y"], r["grid_z"])] = (int(r["calls"]), float(r["avg_us"]), float(r["total_us"]))
    return d
a, b = load("gpurun_out/r06_c23_f32_bpermute_kernels.csv"), load("gpurun_out/r06_c23_f32_new_kernels.csv")
rows = []
for k in a:
    if k in b:
        rows.append((b[k][2] - a[k][2], k, a[k], b[k]))
rows.sort(key=lambda r: -abs(r[0]))
for d, k, x, y in rows[:14]:
    print("%+9.1f us total  %s grid %s: %d calls, %.2f -> %.2f us" % (d, k[0][:60], k[1:], x[0], x[1], y[1]))
PY
