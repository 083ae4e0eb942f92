#!/bin/bash
# usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r2_call2.sh'
# where does the time of the dec GEMM go?  ablations + tilings under rocprofv3 kernel trace; then HBM / L2 counters of the bench
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "config,kernel,grid,calls,avg_us" > $R/gpurun_out/dec_ablate.csv
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
    with open('gpurun_out/dec_ablate.csv', 'a') as f:
        for r in rows:
            nm = r[0].split('::')[-1][:40]
            f.write('%s,"%s",%d,%d,%.3f\n' % (label, nm, r[1], r[2], r[3]))
PY
}
abl base SWX_DEC_ABL=0
abl no_weights SWX_DEC_ABL=1
abl no_adma SWX_DEC_ABL=2
abl no_w_no_a SWX_DEC_ABL=3
abl no_stats SWX_DEC_ABL=4
abl no_mfma SWX_DEC_ABL=8
abl no_epi SWX_DEC_ABL=16
abl only_launch SWX_DEC_ABL=31
abl only_weights SWX_DEC_ABL=30
abl spread_xcd SWX_DEC_ABL=32
abl mt2_small SWX_DEC_ABL=0 SWX_DEC_POLICY="1280x1280=2:1,3840x1280=3:1,5120x1280=2:1,1280x5120=2:4"
abl mt3_small SWX_DEC_ABL=0 SWX_DEC_POLICY="1280x1280=3:1,3840x1280=1:1,5120x1280=1:1,1280x5120=1:4"
abl ks8 SWX_DEC_ABL=0 SWX_DEC_POLICY="1280x5120=2:8"
abl m50 SWX_DEC_ABL=0 DEC_M=50
abl m16 SWX_DEC_ABL=0 DEC_M=16
cat gpurun_out/dec_ablate.csv
# counters of a shortened bench pass (8 decode steps; separate passes, kernel-trace only)
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $ctr | tr ' ' '_')
  rm -rf /tmp/pmc_prof; cd /tmp
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_prof -o pmc -- python $R/bench.py --steps 1 --warmup 0 --tokens 8 --no-cpu-baseline --no-roofline --no-f32 > /tmp/pmc.log 2>&1 || tail -3 /tmp/pmc.log
  cd $R
  python - "$tag" <<'PY'
import sqlite3, glob, sys
tag = sys.argv[1]
out = open('gpurun_out/pmc_%s.csv' % tag, 'w')
for db in glob.glob('/tmp/pmc_prof/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    try:
        cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
        namecol = 'kernel_name' if 'kernel_name' in cols else ('name' if 'name' in cols else cols[0])
        cn = 'counter_name' if 'counter_name' in cols else 'pmc_name'
        val = 'value' if 'value' in cols else 'counter_value'
        gs = 'grid_size' if 'grid_size' in cols else '0'
        out.write("kernel,grid_size,counter,calls,avg,sum\n")
        q = "select %s, %s, %s, count(*), avg(%s), sum(%s) from counters_collection group by %s, %s, %s order by sum(%s) desc limit 60" % (namecol, gs, cn, val, val, namecol, gs, cn, val)
        for r in c.execute(q):
            out.write('"%s",%s,%s,%d,%.3f,%.1f\n' % (str(r[0]).split('::')[-1][:60], r[1], r[2], r[3], r[4], r[5]))
    except Exception as e:
        out.write("# error %r\n" % (e,))
out.close()
PY
  head -14 gpurun_out/pmc_$tag.csv
done
# strict f32 leg alone (the first call reported an asynchronous HIP error there)
( AMD_SERIALIZE_KERNEL=3 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline 2>&1 | tail -4 ) > gpurun_out/bench_f32leg.log; cut -c1-1500 gpurun_out/bench_f32leg.log
# new DTW kernel + remaining GPU tests that the first call's -x cut off
( timeout 600 python -m pytest tests -m gpu -q -n 4 --timeout=600 --deselect tests/test_gpu_largev3.py::test_lv3_full_depth_single_window_greedy_f32 2>&1 | tail -25 ) > gpurun_out/gpu_tests2.log; cat gpurun_out/gpu_tests2.log
( timeout 120 python scripts/kernel_bench.py --only dtw 2>&1 | tail -8 ) > gpurun_out/kb_dtw.log; cat gpurun_out/kb_dtw.log
