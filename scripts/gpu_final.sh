#!/bin/bash
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_final.sh [tag]'
# The round's measurement set: (1) HBM traffic counters per kernel class (separate FETCH_SIZE / WRITE_SIZE passes, kernel-trace
# only) -> profiles/pmc_traffic.json (read by bench.py's roofline object), (2) rocprofv3 --kernel-trace --stats of the default
# bench command, (3) the default bench line (with the strict-f32 leg and the CPU baseline), (4) smoke + the GPU test suite.
tag=${1:-final}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  ( timeout 500 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --tokens 8 --no-cpu-baseline --no-roofline --no-f32 --debug-flags 16384 2>&1 | tail -2 ) > $R/gpurun_out/pmc_$C.log
done
cd $R
python - "$tag" <<'PY'
import sqlite3, glob, json, sys, collections
tag = sys.argv[1]
CLASSES = [("gemm_dec_f16", "decode-step GEMM"), ("gemm_dectall_f16", "decode-step GEMM"), ("attn_decode_cross", "attn_decode_cross_f16"), ("gemm_f16_glds", "gemm_f16_tiled"), ("gemm_f16_big", "gemm_f16_tiled"), ("gemm_f16_ring", "gemm_f16_tiled"),
           ("gemm_f16_tiled", "gemm_f16_tiled"), ("attn_flash", "attn_flash_f16"), ("self_attn_step", "self_attn (decode step)"),
           ("decode_select", "decode_select"), ("dec_slab_finish", "splitk_finish / layernorm"), ("layernorm_kernel", "splitk_finish / layernorm"),
           ("swx_dtw", "dtw"), ("swx_align", "align_weights"), ("swx_mel", "mel")]
per_kernel = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for db in glob.glob(f'/tmp/pmc_{C}/**/*.db', recursive=True):
        c = sqlite3.connect(db)
        cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
        namecol = 'kernel_name' if 'kernel_name' in cols else 'name'
        cn = 'counter_name' if 'counter_name' in cols else 'pmc_name'
        val = 'value' if 'value' in cols else 'counter_value'
        grid = 'grid_size' if 'grid_size' in cols else None
        q = f"select {namecol}, {grid or 0}, count(*), sum({val}) from counters_collection where {cn} = '{C}' group by {namecol}, {grid or 0}"
        for name, g, n, tot in c.execute(q):
            k = (str(name)[:80], int(g))
            per_kernel[k][C][0] += float(tot)
            per_kernel[k][C][1] += int(n)
with open(f'gpurun_out/pmc_{tag}.csv', 'w') as f:
    f.write("kernel,grid_size,launches,FETCH_SIZE_KB_avg(raw),WRITE_SIZE_KB_avg,bytes_per_launch(2*fetch+write)\n")
    rows = []
    for (name, g), d in per_kernel.items():
        nf, nw = d["FETCH_SIZE"][1], d["WRITE_SIZE"][1]
        fa = d["FETCH_SIZE"][0] / nf if nf else 0.0
        wa = d["WRITE_SIZE"][0] / nw if nw else 0.0
        rows.append((name, g, max(nf, nw), fa, wa, (2 * fa + wa) * 1024))
    rows.sort(key=lambda r: -r[5] * r[2])
    for r in rows[:60]:
        f.write('"%s",%d,%d,%.3f,%.3f,%.0f\n' % r)
agg = collections.defaultdict(lambda: [0.0, 0])
for name, g, n, fa, wa, b in rows:
    for key, cls in CLASSES:
        if key in name:
            agg[cls][0] += b * n
            agg[cls][1] += n
            break
out = {cls: {"bytes_per_launch": round(t / n), "launches_sampled": n,
             "source": f"profiles/r04_pmc_{tag}.csv: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate kernel-trace passes, "
                       "8 decode steps of the bench workload); FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, "
                       "MI355X_MICROARCH.md), KB -> bytes"} for cls, (t, n) in agg.items() if n}
json.dump(out, open('gpurun_out/pmc_traffic.json', 'w'), indent=1)
print(json.dumps({k: v["bytes_per_launch"] for k, v in out.items()}))
PY
rm -rf /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
mkdir -p profiles && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json     # so that the bench line below carries `traffic`
scripts/rocprof_kernels.sh bench_$tag python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
head -30 gpurun_out/bench_${tag}_kernels.csv
echo "== smoke"; ( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 )
echo "== default bench line"; ( timeout 900 python bench.py 2>&1 | tail -1 ) | tee gpurun_out/bench_$tag.json | cut -c1-4000
if [ -z "$SKIP_SUITE" ]; then echo "== gpu suite"; ( timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -8 ) | tee gpurun_out/gpu_suite_$tag.log; fi
