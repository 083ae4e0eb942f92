#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -n 2 --timeout=300 2>&1 | tail -30 ) > gpurun_out/kernels.log
( timeout 420 python -m pytest tests/test_gpu_golden.py "tests/test_gpu_model.py::test_decode_f16_fast_step_equals_general_path" -m gpu -q -n 3 --timeout=400 2>&1 | tail -30 ) > gpurun_out/model.log
( timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/bench.err | tail -5 ) > gpurun_out/bench.log
( SWX_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 1 --warmup 0 --minutes 1 --batch 2 --no-cpu-baseline --no-roofline 2>&1 | tail -4 ) > gpurun_out/dist1.log
cd /tmp && ( timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
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
tail -6 gpurun_out/kernels.log; cat gpurun_out/dist1.log; tail -12 gpurun_out/model.log; tail -4 gpurun_out/bench.err; cat gpurun_out/bench.log; head -12 gpurun_out/kernel_stats.csv; head -30 gpurun_out/kernel_by_grid.csv
