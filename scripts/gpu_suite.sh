#!/bin/bash
# usage: gpurun -- 'bash scripts/gpu_suite.sh'
# whole GPU suite, smoke, bench line, rocprofv3 kernel stats of the bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 60 python __graft_entry__.py smoke 2>&1 | tail -3 ) > gpurun_out/smoke.log
( timeout 900 python -m pytest tests -m gpu -q -n 4 --timeout=400 2>&1 | tail -8 ) > gpurun_out/gpu_tests.log
( timeout 240 python bench.py 2> gpurun_out/bench.err | tail -2 ) > gpurun_out/bench.log
cd /tmp && ( timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
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
    rows = c.execute("select name, grid_x, grid_y, grid_z, count(*), avg(duration)/1000.0, sum(duration)/1000.0 from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc limit 40").fetchall()
    with open('gpurun_out/kernel_by_grid.csv', 'w') as f:
        f.write("name,grid_x,grid_y,grid_z,calls,avg_us,total_us\n")
        for r in rows:
            f.write('"%s",%d,%d,%d,%d,%.3f,%.3f\n' % (r[0][:60], r[1], r[2], r[3], r[4], r[5], r[6]))
PY
rm -f gpurun_out/prof/*.db
cat gpurun_out/smoke.log; tail -5 gpurun_out/gpu_tests.log; tail -2 gpurun_out/bench.err; cut -c1-1800 gpurun_out/bench.log; head -14 gpurun_out/kernel_stats.csv
