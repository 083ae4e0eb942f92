#!/bin/bash
# round 6 call 36: where the headline pass's word-timestamp stage goes after the scoring-pass kernels got faster: stage times (a synchronise per
# stage) and a host profile of one pass
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --phase-times --host-profile gpurun_out/r06_c36_host_profile.txt > gpurun_out/r06_c36_bench_phase.json 2> gpurun_out/r06_c36.err )
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_c36_bench_phase.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
print(json.dumps(d.get("phase_times_ms") or d.get("phase_times") or {k: v for k, v in d.items() if "phase" in k}, indent=0)[:3000])
PY
head -60 gpurun_out/r06_c36_host_profile.txt | cut -c1-200
tail -3 gpurun_out/r06_c36.err
