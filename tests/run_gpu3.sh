#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -n 2 --timeout=600 2>&1 | tail -40 ) > gpurun_out/kernels.log
( timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py -m gpu -q -n 3 --timeout=900 2>&1 | tail -60 ) > gpurun_out/model.log
( timeout 900 python bench.py 2> gpurun_out/bench.err | tail -5 ) > gpurun_out/bench.log
cd /tmp && ( timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
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
tail -8 gpurun_out/kernels.log; tail -25 gpurun_out/model.log; tail -30 gpurun_out/bench.err; cat gpurun_out/bench.log; head -14 gpurun_out/kernel_stats.csv
