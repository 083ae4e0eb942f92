#!/bin/bash
# usage: scripts/rocprof_kernels.sh <tag> <command ...>   -> gpurun_out/<tag>_kernels.csv (name, grid, calls, avg / min / total us)
# rocprofv3 --kernel-trace of the command; the sqlite output is reduced to one line per (kernel, grid) and removed.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o t -- "$@" > $R/gpurun_out/${tag}_cmd.log 2>&1 )
python - "$tag" "$R" <<'PY'
import sqlite3, glob, sys
tag, R = sys.argv[1], sys.argv[2]
for db in glob.glob(f'/tmp/prof_{tag}/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    rows = c.execute("select name, grid_x, grid_y, grid_z, count(*), avg(duration)/1000.0, min(duration)/1000.0, sum(duration)/1000.0 "
                     "from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc limit 60").fetchall()
    with open(f'{R}/gpurun_out/{tag}_kernels.csv', 'w') as f:
        f.write("name,grid_x,grid_y,grid_z,calls,avg_us,min_us,total_us\n")
        for r in rows:
            f.write('"%s",%d,%d,%d,%d,%.3f,%.3f,%.3f\n' % (r[0][:90], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
    # idle time in front of each kernel (start - end of the previous kernel on the device), summed per kernel name
    ks = c.execute("select name, start, end from kernels order by start").fetchall()
    import collections
    gap = collections.defaultdict(lambda: [0, 0.0, 0.0])
    busy, t_first, t_last, prev_end = 0.0, ks[0][1], ks[-1][2], None
    for name, st, en in ks:
        busy += (en - st) / 1000.0
        if prev_end is not None:
            g = max(0.0, (st - prev_end) / 1000.0)
            e = gap[name[:70]]
            e[0] += 1; e[1] += g; e[2] = max(e[2], g)
        prev_end = max(prev_end or en, en)
    lo = len(ks) * 2 // 3
    with open(f'{R}/gpurun_out/{tag}_slice.csv', 'w') as f:          # a slice of the timeline, two thirds into the run
        f.write("idx,name,start_us_rel,dur_us,gap_before_us\n")
        t0s = ks[lo][1]
        for q in range(lo, min(len(ks), lo + 700)):
            name, st, en = ks[q]
            f.write('%d,"%s",%.1f,%.1f,%.1f\n' % (q, name[:60], (st - t0s) / 1000.0, (en - st) / 1000.0, (st - ks[q - 1][2]) / 1000.0))
    with open(f'{R}/gpurun_out/{tag}_gaps.csv', 'w') as f:
        f.write("# wall %.1f ms, kernels busy %.1f ms, idle %.1f ms\n" % ((t_last - t_first) / 1e6, busy / 1e3, (t_last - t_first) / 1e6 - busy / 1e3))
        f.write("kernel_after_gap,count,total_gap_us,avg_gap_us,max_gap_us\n")
        for name, (n, tot, mx) in sorted(gap.items(), key=lambda kv: -kv[1][1])[:40]:
            f.write('"%s",%d,%.1f,%.2f,%.1f\n' % (name, n, tot, tot / n, mx))
PY
rm -rf /tmp/prof_$tag
