#!/bin/bash
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r2_call5.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q -n 4 --timeout=600 --deselect tests/test_gpu_largev3.py::test_lv3_full_depth_single_window_greedy_f32 2>&1 | tail -30 ) > gpurun_out/gpu_tests5.log; tail -30 gpurun_out/gpu_tests5.log
for cfg in "SWX_DTW_ABL=0" "SWX_DTW_ABL=1"; do
  echo "== dtw $cfg"; ( env $cfg timeout 100 python scripts/kernel_bench.py --only dtw 2>&1 | grep "W=" )
done
( timeout 100 python scripts/kernel_bench.py --only flash 2>&1 | tail -3 ); ( SWX_FLAGS=$((4|16|64|512|256|4096)) timeout 100 python scripts/kernel_bench.py --only flash 2>&1 | tail -2 )
( timeout 100 python scripts/kernel_bench.py --only cross 2>&1 | tail -2 )
( timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 2>&1 | tail -1 ) > gpurun_out/bench5.log; cut -c1-2800 gpurun_out/bench5.log
( SWX_FLAGS=$((4|16|64|512|256|2048)) timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-400 ) | tee gpurun_out/bench5_nopack.log
( SWX_FLAGS=$((4|16|64|512|256|4096)) timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-400 ) | tee gpurun_out/bench5_flashv1.log
( SWX_FLAGS=$((4|16|64|512)) timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-400 ) | tee gpurun_out/bench5_noglds.log
cd /tmp && ( timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 2>&1 | tail -3 ) > $R/gpurun_out/rocprof.log
cd $R
python - <<'PY'
import sqlite3, glob
for db in glob.glob('gpurun_out/prof/*.db'):
    c = sqlite3.connect(db)
    rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    with open('gpurun_out/kernel_stats.csv', 'w') as f:
        f.write("name,calls,total_us,avg_us,percent\n")
        for r in rows:
            f.write('"%s",%d,%.3f,%.3f,%.4f\n' % r)
    rows = c.execute("select name, grid_x, grid_y, grid_z, count(*), avg(duration)/1000.0, sum(duration)/1000.0 from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc limit 50").fetchall()
    with open('gpurun_out/kernel_by_grid.csv', 'w') as f:
        f.write("name,grid_x,grid_y,grid_z,calls,avg_us,total_us\n")
        for r in rows:
            f.write('"%s",%d,%d,%d,%d,%.3f,%.3f\n' % (r[0][:70], r[1], r[2], r[3], r[4], r[5], r[6]))
PY
rm -f gpurun_out/prof/*.db
head -22 gpurun_out/kernel_stats.csv
( timeout 300 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 2>&1 | tail -1 | cut -c1-1800 ) | tee gpurun_out/bench5_align.log
