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
PY
rm -rf /tmp/prof_$tag
