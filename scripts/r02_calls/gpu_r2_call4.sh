#!/bin/bash
# usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r2_call4.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_largev3.py -m gpu -q -n 4 --timeout=300 -k "dec_gemm or dec_step or dtw or f16_decode" 2>&1 | tail -5 ) > gpurun_out/gpu_tests4.log; cat gpurun_out/gpu_tests4.log
echo "config,kernel,grid,calls,avg_us" > $R/gpurun_out/dec_ablate3.csv
abl() {  # label, env assignments...
  label=$1; shift
  rm -rf /tmp/abl_prof; cd /tmp
  env "$@" timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/abl_prof -o abl -- python $R/scripts/dec_ablate.py > /tmp/abl.log 2>&1 || tail -3 /tmp/abl.log
  cd $R
  python - "$label" <<'PY'
import sqlite3, glob, sys
label = sys.argv[1]
for db in glob.glob('/tmp/abl_prof/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    rows = c.execute("select name, grid_x, count(*), avg(duration)/1000.0 from kernels where name like '%gemm_dec%' or name like '%slab_finish%' group by name, grid_x order by name, grid_x").fetchall()
    with open('gpurun_out/dec_ablate3.csv', 'a') as f:
        for r in rows:
            nm = r[0].split('::')[-1][:44]
            f.write('%s,"%s",%d,%d,%.3f\n' % (label, nm, r[1], r[2], r[3]))
PY
}
abl packed SWX_DEC_ABL=0
abl packed_mt2 SWX_DEC_ABL=0 SWX_DEC_POLICY="1280x1280=2:1,3840x1280=3:1,5120x1280=2:1,1280x5120=2:4"
abl packed_mt1 SWX_DEC_ABL=0 SWX_DEC_POLICY="3840x1280=1:1,5120x1280=1:1,1280x5120=1:4"
abl packed_ks8 SWX_DEC_ABL=0 SWX_DEC_POLICY="1280x5120=1:8"
abl packed_m50 SWX_DEC_ABL=0 DEC_M=50
cat gpurun_out/dec_ablate3.csv
for cfg in "SWX_DTW_ABL=0" "SWX_DTW_ABL=1"; do
  echo "== dtw $cfg"; ( env $cfg timeout 100 python scripts/kernel_bench.py --only dtw 2>&1 | grep "W=" ) | tee -a gpurun_out/kb_dtw3.log
done
( timeout 400 python bench.py --steps 3 --warmup 1 2>&1 | tail -2 ) > gpurun_out/bench4.log; cut -c1-3000 gpurun_out/bench4.log
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
head -16 gpurun_out/kernel_stats.csv
