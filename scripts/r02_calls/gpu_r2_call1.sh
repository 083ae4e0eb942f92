#!/bin/bash
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r2_call1.sh'
# round 2, first hardware call: whole GPU suite (incl. the new large-v3-dims parity tests and the dec GEMM tests), kernel
# micro-benchmarks old vs new decode GEMM, the bench line on the new workload (old step vs dec step), streams / spans / align modes
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout ${T:-240} "$@" > /tmp/swx_step.log 2>&1; rc=$?; tail -${L:-25} /tmp/swx_step.log > gpurun_out/$name.log; echo "== $name: exit $rc"; cat gpurun_out/$name.log; }
L=4 T=120 run smoke python __graft_entry__.py smoke
L=60 T=900 run gpu_tests python -m pytest tests -m gpu -q -n 4 --timeout=600 -x --deselect tests/test_gpu_largev3.py::test_lv3_full_depth_single_window_greedy_f32
L=30 T=400 run gpu_tests_full_depth python -m pytest tests/test_gpu_largev3.py -m gpu -q -k full_depth --timeout=380
run kb_dec python scripts/kernel_bench.py --only dec
run kb_splitk python scripts/kernel_bench.py --only splitk
SWX_DEC_POLICY="1280x5120=2:8,5120x1280=2:1,3840x1280=3:1" run kb_dec_p2 python scripts/kernel_bench.py --only dec
SWX_DEC_POLICY="1280x1280=2:1,1280x5120=1:4,5120x1280=1:1,3840x1280=1:1" run kb_dec_p3 python scripts/kernel_bench.py --only dec
run kb_gemm1 python scripts/kernel_bench.py --only gemm --gemm-kernel 1
run kb_gemm4 python scripts/kernel_bench.py --only gemm --gemm-kernel 4
run kb_attn python scripts/kernel_bench.py --only cross
run kb_flash python scripts/kernel_bench.py --only flash
L=6 T=420 run bench_default python bench.py --steps 3 --warmup 1
SWX_FLAGS=84 L=3 T=200 run bench_oldstep python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32
L=3 T=200 run bench_streams2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --streams 2
L=3 T=240 run bench_spans20 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --spans 20
L=3 T=300 run bench_align python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32
cd /tmp && ( timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 2>&1 | tail -3 ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
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
    rows = c.execute("select name, grid_x, grid_y, grid_z, count(*), avg(duration)/1000.0, sum(duration)/1000.0 from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc limit 50").fetchall()
    with open('gpurun_out/kernel_by_grid.csv', 'w') as f:
        f.write("name,grid_x,grid_y,grid_z,calls,avg_us,total_us\n")
        for r in rows:
            f.write('"%s",%d,%d,%d,%d,%.3f,%.3f\n' % (r[0][:70], r[1], r[2], r[3], r[4], r[5], r[6]))
PY
rm -f gpurun_out/prof/*.db
head -30 gpurun_out/kernel_stats.csv
