#!/bin/bash
# usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r2_call3.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -n 4 --timeout=300 -k "dec_gemm or dec_step or dtw" 2>&1 | tail -5 ) > gpurun_out/gpu_tests3.log; cat gpurun_out/gpu_tests3.log
echo "config,kernel,grid,calls,avg_us" > $R/gpurun_out/dec_ablate2.csv
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
    with open('gpurun_out/dec_ablate2.csv', 'a') as f:
        for r in rows:
            nm = r[0].split('::')[-1][:44]
            f.write('%s,"%s",%d,%d,%.3f\n' % (label, nm, r[1], r[2], r[3]))
PY
}
abl overlap SWX_DEC_ABL=0
abl hot SWX_DEC_ABL=0 DEC_HOT=1
abl touch0 SWX_DEC_ABL=0 DEC_TOUCH=0
abl touch12 SWX_DEC_ABL=0 DEC_TOUCH=12
abl touch40 SWX_DEC_ABL=0 DEC_TOUCH=40
cat gpurun_out/dec_ablate2.csv
for cfg in "SWX_DTW_ABL=0" "SWX_DTW_ABL=1" "SWX_DTW_ABL=2" "SWX_DTW_ABL=3" "SWX_DTW_ABL=5" "SWX_DTW_ABL=0 SWX_DTW_CH=32" "SWX_DTW_ABL=1 SWX_DTW_CH=32"; do
  echo "== dtw $cfg"; ( env $cfg timeout 100 python scripts/kernel_bench.py --only dtw 2>&1 | grep "W=" ) | tee -a gpurun_out/kb_dtw2.log
done
SWX_DTW_CH=32 timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k dtw 2>&1 | tail -2
( timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 2>&1 | tail -2 ) > gpurun_out/bench3.log; cut -c1-2600 gpurun_out/bench3.log
( AMD_SERIALIZE_KERNEL=3 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -12 ) > gpurun_out/bench_f32leg2.log; cut -c1-1800 gpurun_out/bench_f32leg2.log
