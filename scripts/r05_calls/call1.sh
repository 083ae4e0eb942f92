#!/bin/bash
# Round 5, GPU call 1 (information only): (1) rocprofv3 kernel summary of the STRICT-F32 pass (none existed: VERDICT r4 weak / item 1e),
# (2) MFMA-utilisation COUNTERS (SQ_VALU_MFMA_BUSY_CYCLES next to GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES) on a shortened f16 pass,
# (3) rocprofv3 kernel summary of align mode and of the sequential mode at HEAD (starting points of items 5 / 6).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
step() { echo "== $1 ($(date +%T))"; }
step "f32 pass kernels"
bash scripts/rocprof_kernels.sh r05_c1_f32pass python $R/bench.py --dtype f32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
tail -2 gpurun_out/r05_c1_f32pass_cmd.log | cut -c1-300; head -30 gpurun_out/r05_c1_f32pass_kernels.csv; head -3 gpurun_out/r05_c1_f32pass_gaps.csv
step "mfma counters"
cd /tmp
( timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace -d /tmp/pmc_mfma -o pmc -- python $R/bench.py --steps 1 --warmup 0 --tokens 8 --no-cpu-baseline --no-roofline --no-f32 2>&1 | tail -2 ) > $R/gpurun_out/r05_c1_pmc_mfma.log
cd $R
python - <<'PY'
import sqlite3, glob
out = open('gpurun_out/r05_c1_pmc_mfma.csv', 'w')
for db in sorted(glob.glob('/tmp/pmc_mfma/**/*.db', recursive=True)):
    c = sqlite3.connect(db)
    try:
        cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
        out.write("# columns: %s\n" % cols)
        namecol = 'kernel_name' if 'kernel_name' in cols else ('name' if 'name' in cols else cols[0])
        cn = 'counter_name' if 'counter_name' in cols else 'pmc_name'
        val = 'value' if 'value' in cols else 'counter_value'
        q = "select %s, %s, count(*), avg(%s), sum(%s) from counters_collection group by %s, %s order by %s, %s" % (namecol, cn, val, val, namecol, cn, namecol, cn)
        for r in c.execute(q):
            out.write('"%s",%s,%d,%.3f,%.3f\n' % (str(r[0])[:80], r[1], r[2], r[3], r[4]))
    except Exception as e:
        out.write("# error %r\n" % (e,))
out.close()
PY
tail -2 gpurun_out/r05_c1_pmc_mfma.log | cut -c1-200; grep -i "big8\|flash2\|glds_128" gpurun_out/r05_c1_pmc_mfma.csv | cut -c1-200 | head -30
rm -rf /tmp/pmc_mfma
step "align kernels"
bash scripts/rocprof_kernels.sh r05_c1_align python $R/bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
tail -1 gpurun_out/r05_c1_align_cmd.log | cut -c1-300; head -24 gpurun_out/r05_c1_align_kernels.csv; head -3 gpurun_out/r05_c1_align_gaps.csv
step "sequential kernels"
bash scripts/rocprof_kernels.sh r05_c1_seq python $R/bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
tail -1 gpurun_out/r05_c1_seq_cmd.log | cut -c1-300; head -24 gpurun_out/r05_c1_seq_kernels.csv; head -3 gpurun_out/r05_c1_seq_gaps.csv
step done
