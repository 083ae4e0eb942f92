#!/bin/bash
# round 6 call 29: (a) self_attn_step_long_f16 -- the long-context decode-step self-attention of launches with few waves (sequential flow:
# 100 waves, positions 228-340) with every load of a row requested in two batches instead of 13 dependent round trips (flag 4 =
# SWX_FLAG_SELFATTN_NO_DEEP puts the chunk-by-chunk kernel back): bit-identity check on hardware, A/B in the sequential mode;
# (b) second form of the four-blocks-in-flight cross-attention (extra blocks requested BEHIND the query projection's tile barrier; the first
# form, in front of it, measured 0.7 % slower in the sequential mode: call 28): A/B again (flag 2)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python tests/hw_checks/self_attn_step_check.py 2>&1 | tail -20 ) > gpurun_out/r06_c29_self_attn_check.txt; cat gpurun_out/r06_c29_self_attn_check.txt
( timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_golden.py -m gpu -q -x 2>&1 | tail -5 ) > gpurun_out/r06_c29_tests.log; cat gpurun_out/r06_c29_tests.log
( timeout 900 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 4 > gpurun_out/r06_c29_bench_seq_selfattn_deep_ab.json 2> gpurun_out/r06_c29.err )
( timeout 900 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 2 > gpurun_out/r06_c29_bench_seq_xattn_deep2_ab.json 2>> gpurun_out/r06_c29.err )
( timeout 600 python bench.py --mode align --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 2 > gpurun_out/r06_c29_bench_align_xattn_deep2_ab.json 2>> gpurun_out/r06_c29.err )
cd /tmp && rm -rf /tmp/seqprof && ( timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/seqprof -o seq -- python $GRAFT_REPO_ROOT/bench.py --sequential --minutes 2 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline > /dev/null 2>> $GRAFT_REPO_ROOT/gpurun_out/r06_c29.err ); cd $GRAFT_REPO_ROOT
f=$(find /tmp/seqprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" > gpurun_out/r06_c29_seq_kernel_stats.csv
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c29_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c29_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
head -16 gpurun_out/r06_c29_seq_kernel_stats.csv | cut -c1-160
tail -3 gpurun_out/r06_c29.err
