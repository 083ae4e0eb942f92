#!/bin/bash
# usage: gpurun --timeout 1800 -- 'bash scripts/r06_calls/call47.sh'
# Last call of round 6 on the final tree, in the driver's order: (1) batch invariance swept over ALL 20 windows of the timed batch
# (f16 and strict f32), (2) the serial GPU suite as the driver runs it, (3) smoke, (4) the default bench line with the driver's step
# counts, (5) the same bench through torch.distributed.run at one rank (the launch path of --gpus N).
mkdir -p gpurun_out
export TMPDIR=/tmp
( SWX_BENCH_WINDOWS=all timeout 900 python -m pytest tests/test_gpu_batch_invariance.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/r06_c47_batch_invariance_all20.log
for dt in f16 f32; do cp gpurun_out/batch_invariance_report_$dt.json gpurun_out/r06_c47_batch_invariance_all20_$dt.json 2>/dev/null; done
tail -3 gpurun_out/r06_c47_batch_invariance_all20.log
( timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r06_final13_gpu_suite.log; tail -2 gpurun_out/r06_final13_gpu_suite.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/r06_final13_smoke.log; cat gpurun_out/r06_final13_smoke.log
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r06_final13_bench.json 2> gpurun_out/r06_final13_bench.err ); tail -2 gpurun_out/r06_final13_bench.err
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-f32 > gpurun_out/r06_final13_bench_torchrun_n1.json 2> gpurun_out/r06_final13_bench_torchrun_n1.err ); tail -2 gpurun_out/r06_final13_bench_torchrun_n1.err
python - <<'PY'
import json
for f in ("gpurun_out/r06_final13_bench.json", "gpurun_out/r06_final13_bench_torchrun_n1.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["n_gpus"], d["config"]["words"], (d.get("roofline") or {}).get("frac"), (d.get("strict_f32") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable", e)
PY
