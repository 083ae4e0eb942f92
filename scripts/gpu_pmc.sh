#!/bin/bash
# usage: gpurun -- 'bash scripts/gpu_pmc.sh'
# HBM traffic counters (separate passes, kernel-trace only) on a shortened decode (8 steps) of the bench workload
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  ( timeout 500 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --tokens 8 --no-cpu-baseline --no-roofline 2>&1 | tail -2 ) > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob
out = open('gpurun_out/pmc_summary.csv', 'w')
for db in sorted(glob.glob('gpurun_out/pmc_*/*.db')):
    c = sqlite3.connect(db)
    try:
        cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
        out.write("# %s columns: %s\n" % (db, cols))
        namecol = 'kernel_name' if 'kernel_name' in cols else ('name' if 'name' in cols else cols[0])
        cn = 'counter_name' if 'counter_name' in cols else 'pmc_name'
        val = 'value' if 'value' in cols else 'counter_value'
        q = "select %s, %s, count(*), avg(%s), sum(%s) from counters_collection group by %s, %s order by sum(%s) desc limit 25" % (namecol, cn, val, val, namecol, cn, val)
        for r in c.execute(q):
            out.write('"%s",%s,%d,%.3f,%.3f\n' % (str(r[0])[:70], r[1], r[2], r[3], r[4]))
    except Exception as e:
        out.write("# error %s: %r\n" % (db, e))
        for t in c.execute("select name from sqlite_master where type in ('table','view') and name like '%pmc%'"):
            out.write("# table %s: %s\n" % (t[0], [d[1] for d in c.execute('pragma table_info(%s)' % t[0])]))
out.close()
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cat gpurun_out/pmc_summary.csv | cut -c1-200; tail -2 gpurun_out/pmc_FETCH_SIZE.log
