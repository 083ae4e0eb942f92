#!/bin/bash
# round 6 call 16: block reductions of the selection kernels (16 wave slots combined by a lane butterfly instead of a walk by every thread):
# decode tests, A/B of two builds on one box (scripts/exp/libswx_r5reduce.so = -DSWX_SELECT_R5_REDUCE), kernel summary; SQ counters of the
# decode-step kernels on a shortened decode (one counter per pass, kernel-trace only)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
( timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r06_c16_tests.log
cat gpurun_out/r06_c16_tests.log
cp stable_ts_amd/libswx.so /tmp/libswx_new.so
for lib in r5reduce new r5reduce new; do
    if [ $lib = r5reduce ]; then cp scripts/exp/libswx_r5reduce.so stable_ts_amd/libswx.so; else cp /tmp/libswx_new.so stable_ts_amd/libswx.so; fi
    ( timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline ) >> gpurun_out/r06_c16_bench_${lib}.txt 2>> gpurun_out/r06_c16_bench.err
done
cp /tmp/libswx_new.so stable_ts_amd/libswx.so
python - <<'PY'
import json
for l in ("r5reduce", "new"):
    rows = [json.loads(x) for x in open(f"gpurun_out/r06_c16_bench_{l}.txt") if x.startswith("{")]
    print("bench", l, [r["ms_per_step"] for r in rows], [r["config"].get("words") for r in rows])
PY
bash scripts/rocprof_kernels.sh r06_c16_pass python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline
grep -i "select\|beam\|finish\|self_attn_step\|embed" gpurun_out/r06_c16_pass_kernels.csv | cut -c1-150
cd /tmp
for C in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE; do
  ( timeout 200 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/sq_$C -o pmc -- python $R/bench.py --steps 1 --warmup 0 --tokens 8 --no-cpu-baseline --no-roofline --no-f32 2>&1 | tail -2 ) > $R/gpurun_out/sq_$C.log
done
cd $R
python - <<'PY'
import sqlite3, glob, json, collections
res = collections.defaultdict(dict)
for db in sorted(glob.glob('gpurun_out/sq_*/*.db')):
    c = sqlite3.connect(db)
    try:
        cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
        namecol = 'kernel_name' if 'kernel_name' in cols else ('name' if 'name' in cols else cols[0])
        cn = 'counter_name' if 'counter_name' in cols else 'pmc_name'
        val = 'value' if 'value' in cols else 'counter_value'
        for r in c.execute(f"select {namecol}, {cn}, count(*), avg({val}) from counters_collection group by {namecol}, {cn}"):
            k = str(r[0])[:80]
            if any(t in k for t in ("select", "self_attn_step", "cross_xq", "gemm_dec_f16", "slab_finish", "beam_update", "flash2", "big8")):
                res[k][r[1]] = {"launches": r[2], "avg": r[3]}
    except Exception as e:
        res["error " + db] = {"e": repr(e)}
json.dump(res, open('gpurun_out/r06_c16_sq_counters.json', 'w'), indent=1)
for k, v in res.items():
    print(k[:70], {a: round(b.get("avg", 0)) for a, b in v.items() if isinstance(b, dict) and "avg" in b})
PY
rm -rf gpurun_out/sq_*/
