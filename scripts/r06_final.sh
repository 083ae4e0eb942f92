#!/bin/bash
# usage: gpurun --timeout 1800 -- 'bash scripts/r06_final.sh [tag]'
# Round 6's measurement set on the final tree:
#  (1) HBM traffic per kernel class: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE kernel-trace-only passes (8 decode
#      steps of the bench workload, eager launches) -> profiles/pmc_traffic.json, read by bench.py's roofline object;
#  (2) MFMA-utilisation COUNTERS (north star; VERDICT r4 missing 5): SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F16 /_F32,
#      GRBM_GUI_ACTIVE in one more pass -> gpurun_out/r06_<tag>_mfma_util.json: utilisation = MFMA busy cycles / (1024 SIMDs x the
#      kernel's cycles), cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs: checked against the kernel's duration);
#  (3) rocprofv3 --kernel-trace summary of the default bench command; (4) smoke + the complete default bench line.
tag=${1:-final}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
step() { echo "== $1 ($(date +%T))"; }
BENCH8="python $R/bench.py --steps 1 --warmup 0 --tokens 8 --no-cpu-baseline --no-roofline --no-f32 --debug-flags 16384"
cd /tmp
step "pmc passes"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  ( timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o pmc -- $BENCH8 2>&1 | tail -2 ) > $R/gpurun_out/pmc_$C.log
done
# (1b) the same two traffic passes with gemm_f16_big8 on the GROUPED tile order (SWX_FLAG_BIG8_GROUPED = 4194304): does the renumbering cut
#      the fabric traffic of the N = 5120 launch as designed (it does not make the launch faster: profiles/r06_c2_kb_gemm_big_tile_order.txt)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcg_$C
  ( timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmcg_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --tokens 8 --no-cpu-baseline --no-roofline --no-f32 --debug-flags $((16384 + 4194304)) 2>&1 | tail -2 ) > $R/gpurun_out/pmcg_$C.log
done
rm -rf /tmp/pmc_MFMA
( timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_MFMA -o pmc -- $BENCH8 2>&1 | tail -2 ) > $R/gpurun_out/pmc_MFMA.log
( timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_MFMA32 -o pmc -- python $R/bench.py --dtype f32 --steps 1 --warmup 0 --tokens 8 --no-cpu-baseline --no-roofline --no-f32 --debug-flags 16384 2>&1 | tail -2 ) > $R/gpurun_out/pmc_MFMA32.log
cd $R
python - "$tag" <<'PY'
import sqlite3, glob, json, sys, collections
tag = sys.argv[1]
def rows_of(dirname):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for db in glob.glob(f'/tmp/{dirname}/**/*.db', recursive=True):
        c = sqlite3.connect(db)
        cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
        namecol = 'kernel_name' if 'kernel_name' in cols else 'name'
        cn = 'counter_name' if 'counter_name' in cols else 'pmc_name'
        val = 'value' if 'value' in cols else 'counter_value'
        grid = 'grid_size' if 'grid_size' in cols else '0'
        for name, g, ctr, n, tot in c.execute(f"select {namecol}, {grid}, {cn}, count(*), sum({val}) from counters_collection group by {namecol}, {grid}, {cn}"):
            e = out[(str(name)[:90], int(g))][ctr]
            e[0] += float(tot); e[1] += int(n)
    return out
# ---- (1) traffic
CLASSES = [("gemm_dec_f16", "decode-step GEMM"), ("gemm_dectall_f16", "decode-step GEMM"), ("attn_decode_cross", "attn_decode_cross_f16"), ("gemm_f16_glds", "gemm_f16_tiled"), ("gemm_f16_big", "gemm_f16_tiled"), ("gemm_f16_ring", "gemm_f16_tiled"),
           ("gemm_f16_tiled", "gemm_f16_tiled"), ("attn_flash", "attn_flash_f16"), ("self_attn_step", "self_attn (decode step)"),
           ("decode_select", "decode_select"), ("dec_slab_finish", "splitk_finish / layernorm"), ("layernorm_kernel", "splitk_finish / layernorm"),
           ("swx_dtw", "dtw"), ("swx_align", "align_weights"), ("swx_mel", "mel")]
per = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, d in rows_of("pmc_" + C).items():
        if C in d:
            per[k][C][0] += d[C][0]; per[k][C][1] += d[C][1]
rows = []
for (name, g), d in per.items():
    nf, nw = d["FETCH_SIZE"][1], d["WRITE_SIZE"][1]
    fa = d["FETCH_SIZE"][0] / nf if nf else 0.0
    wa = d["WRITE_SIZE"][0] / nw if nw else 0.0
    rows.append((name, g, max(nf, nw), fa, wa, (2 * fa + wa) * 1024))
rows.sort(key=lambda r: -r[5] * r[2])
with open(f'gpurun_out/r06_pmc_{tag}.csv', 'w') as f:
    f.write("kernel,grid_size,launches,FETCH_SIZE_KB_avg(raw),WRITE_SIZE_KB_avg,bytes_per_launch(2*fetch+write)\n")
    for r in rows[:60]:
        f.write('"%s",%d,%d,%.3f,%.3f,%.0f\n' % r)
agg = collections.defaultdict(lambda: [0.0, 0])
for name, g, n, fa, wa, b in rows:
    for key, cls in CLASSES:
        if key in name:
            agg[cls][0] += b * n; agg[cls][1] += n
            break
out = {cls: {"bytes_per_launch": round(t / n), "launches_sampled": n,
             "source": f"profiles/r06_pmc_{tag}.csv: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate kernel-trace passes, "
                       "8 decode steps of the bench workload); FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, "
                       "MI355X_MICROARCH.md), KB -> bytes"} for cls, (t, n) in agg.items() if n}
json.dump(out, open('gpurun_out/pmc_traffic.json', 'w'), indent=1)
# ---- (1b) gemm_f16_big8 under the two tile orders
cmp = {}
for order, pre in (("row-major (default)", "pmc_"), ("grouped (SWX_FLAG_BIG8_GROUPED)", "pmcg_")):
    f_, w_ = rows_of(pre + "FETCH_SIZE"), rows_of(pre + "WRITE_SIZE")
    for k in f_:
        if "gemm_f16_big8" in k[0] and "FETCH_SIZE" in f_[k] and k in w_ and "WRITE_SIZE" in w_[k]:
            fa = f_[k]["FETCH_SIZE"][0] / f_[k]["FETCH_SIZE"][1]; wa = w_[k]["WRITE_SIZE"][0] / w_[k]["WRITE_SIZE"][1]
            cmp.setdefault(f"grid {k[1]}", {})[order] = dict(launches=f_[k]["FETCH_SIZE"][1], fetch_KB_raw=round(fa, 1), write_KB=round(wa, 1),
                                                             bytes_per_launch=round((2 * fa + wa) * 1024))
json.dump(cmp, open(f'gpurun_out/r06_{tag}_big8_traffic_by_tile_order.json', 'w'), indent=1)
print("big8 traffic by tile order:", json.dumps(cmp))
print("traffic:", json.dumps({k: v["bytes_per_launch"] for k, v in out.items()}))
# ---- (2) MFMA utilisation by counters
util = {}
for dirname, mode in (("pmc_MFMA", "f16"), ("pmc_MFMA32", "f32")):
    for (name, g), d in rows_of(dirname).items():
        busy, act = d.get("SQ_VALU_MFMA_BUSY_CYCLES"), d.get("GRBM_GUI_ACTIVE")
        if not busy or not act or busy[0] <= 0 or act[0] <= 0:
            continue
        n = busy[1]
        cycles = act[0] / act[1] / 8.0                       # per launch, per XCD
        mops = sum(d[k][0] / d[k][1] for k in ("SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VALU_MFMA_MOPS_F32") if k in d and d[k][1])
        util[f"{mode}: {name} grid {g}"] = dict(launches=n, mfma_busy_cycles_per_launch=busy[0] / n, kernel_cycles_per_xcd=cycles,
                                                mfma_util=busy[0] / n / (1024.0 * cycles), flops_per_launch=mops * 512.0)
top = dict(sorted(util.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"])[:24])
json.dump(dict(formula="mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), per launch; flops = MFMA_MOPS x 512 "
                       "(counter passes serialise the kernels: cycles include a dispatch tail, so the figure is a lower bound of the in-pass utilisation)",
               kernels=top), open(f'gpurun_out/r06_{tag}_mfma_util.json', 'w'), indent=1)
for k, v in list(top.items())[:12]:
    print("mfma_util %.3f  %s  (%d launches)" % (v["mfma_util"], k[:100], v["launches"]))
PY
rm -rf /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmcg_FETCH_SIZE /tmp/pmcg_WRITE_SIZE /tmp/pmc_MFMA /tmp/pmc_MFMA32
mkdir -p profiles && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json     # so that the bench line below carries `traffic`
step "rocprof pass"
scripts/rocprof_kernels.sh r06_${tag}_pass python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
head -30 gpurun_out/r06_${tag}_pass_kernels.csv; head -3 gpurun_out/r06_${tag}_pass_gaps.csv
step "smoke"; ( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 )
step "default bench line"; ( timeout 600 python bench.py 2>gpurun_out/r06_${tag}_bench.err | tail -1 ) | tee gpurun_out/r06_${tag}_bench.json | cut -c1-3000
step done
