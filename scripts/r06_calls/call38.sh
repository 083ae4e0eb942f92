#!/bin/bash
# round 6 call 38: rocprofv3 kernel tables of the secondary modes on the final tree (sequential, align(), span mode) + stage times of the headline pass
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 bash scripts/rocprof_kernels.sh r06_final8_seq python $R/bench.py --sequential --minutes 2 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) > gpurun_out/r06_c38_rocprof.log 2>&1
( timeout 900 bash scripts/rocprof_kernels.sh r06_final8_align python $R/bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) >> gpurun_out/r06_c38_rocprof.log 2>&1
( timeout 900 bash scripts/rocprof_kernels.sh r06_final8_spans python $R/bench.py --spans 20 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) >> gpurun_out/r06_c38_rocprof.log 2>&1
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --phase-times > gpurun_out/r06_final8_bench_phase_times.json 2> gpurun_out/r06_c38.err )
head -12 gpurun_out/r06_final8_seq_kernels.csv | cut -c1-160; head -2 gpurun_out/r06_final8_seq_gaps.csv
head -14 gpurun_out/r06_final8_align_kernels.csv | cut -c1-160; head -2 gpurun_out/r06_final8_align_gaps.csv
head -8 gpurun_out/r06_final8_spans_kernels.csv | cut -c1-160
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_final8_bench_phase_times.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], json.dumps(d.get("phase_ms"), indent=0))
PY
