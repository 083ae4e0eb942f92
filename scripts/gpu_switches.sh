#!/bin/bash
# usage: gpurun -- 'bash scripts/gpu_switches.sh'
# bit-identity tests of the decode-step switches, timing of the candidate defaults, then the
# bench line and rocprofv3 kernel stats under the fastest bit-identical setting (SWX_FLAGS)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 120 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -n 4 --timeout=100 -k "switches or fast_step or attention or attn" 2>&1 | tail -6 ) > gpurun_out/switch_tests.log
( timeout 100 python scripts/tune_flags.py --flags 20,84,116,20,84 2>&1 | grep "^flags" ) > gpurun_out/tune_flags.log
export SWX_FLAGS=$(cat gpurun_out/best_flags.txt 2>/dev/null || echo 20)
echo "SWX_FLAGS=$SWX_FLAGS" >> gpurun_out/tune_flags.log
( timeout 120 python bench.py 2> gpurun_out/bench.err | tail -2 ) > gpurun_out/bench.log
cd /tmp && ( timeout 100 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob
for db in glob.glob('gpurun_out/prof/*.db'):
    c = sqlite3.connect(db)
    rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    with open('gpurun_out/kernel_stats.csv', 'w') as f:
        f.write("name,calls,total_us,avg_us,percent\n")
        for r in rows:
            f.write('"%s",%d,%.3f,%.3f,%.4f\n' % r)
PY
rm -f gpurun_out/prof/*.db
tail -3 gpurun_out/switch_tests.log; cat gpurun_out/tune_flags.log; cut -c1-700 gpurun_out/bench.log; head -10 gpurun_out/kernel_stats.csv
