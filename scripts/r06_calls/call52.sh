#!/bin/bash
# usage: gpurun --timeout 1800 -- 'bash scripts/r06_calls/call52.sh'
# the round's last tree in the driver's order: serial GPU suite, smoke, default bench line with the driver's step counts
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -22 ) > gpurun_out/r06_final15_gpu_suite.log; tail -3 gpurun_out/r06_final15_gpu_suite.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/r06_final15_smoke.log; cat gpurun_out/r06_final15_smoke.log
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r06_final15_bench.json 2> gpurun_out/r06_final15_bench.err ); tail -1 gpurun_out/r06_final15_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_final15_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["words"], (d.get("roofline") or {}).get("frac"), (d.get("strict_f32") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
PY
