#!/bin/bash
# round 6 call 31: (a) experiment: single-wave dec GEMM workgroups also for the out-projections at 100 rows (140 four-wave workgroups; flag 32 =
# SWX_FLAG_DEC_W1_WIDE): A/B on the headline pass and in the span mode; (b) sequential mode with the cross-attention query projection as
# its own launch (flag 1048576 = SWX_FLAG_NO_FUSED_XQ; the projection now runs as single-wave workgroups): A/B; (c) kernel table of the
# sequential mode on this tree
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 32 > gpurun_out/r06_c31_bench_dec_w1_wide_ab.json 2> gpurun_out/r06_c31.err )
( timeout 600 python bench.py --spans 20 --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 32 > gpurun_out/r06_c31_bench_spans_dec_w1_wide_ab.json 2>> gpurun_out/r06_c31.err )
( timeout 900 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 1048576 > gpurun_out/r06_c31_bench_seq_unfused_xq_ab.json 2>> gpurun_out/r06_c31.err )
( timeout 900 bash scripts/rocprof_kernels.sh r06_c31_seq python $R/bench.py --sequential --minutes 2 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) > gpurun_out/r06_c31_rocprof.log 2>&1
( timeout 900 bash scripts/rocprof_kernels.sh r06_c31_seq_unfused python $R/bench.py --sequential --minutes 2 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --debug-flags 1048576 ) >> gpurun_out/r06_c31_rocprof.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c31_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c31_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
head -16 gpurun_out/r06_c31_seq_kernels.csv | cut -c1-170
head -12 gpurun_out/r06_c31_seq_unfused_kernels.csv | cut -c1-170
tail -3 gpurun_out/r06_c31.err
