#!/bin/bash
# round 6 call 13: lane exchanges on the VALU (v_permlane16/32_swap + DPP instead of ds_bpermute) in the flash kernels' online softmax,
# the decode cross-attention, the wave reductions (token selection: 22 block reductions per row, LayerNorm statistics, self-attention):
# helper check vs __shfl_xor, kernel tests, then the SAME commands on scripts/exp/libswx_bpermute.so (-DSWX_LANE_XOR_BPERMUTE = the
# shuffle form of rounds 1-5) and on the product library, alternating, on one box.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r06_c13_tests.log
cat gpurun_out/r06_c13_tests.log
cp stable_ts_amd/libswx.so /tmp/libswx_new.so
run_pair () {   # $1 = tag, rest = command writing to stdout
    tag=$1; shift
    for lib in bpermute new bpermute new; do
        if [ $lib = bpermute ]; then cp scripts/exp/libswx_bpermute.so stable_ts_amd/libswx.so; else cp /tmp/libswx_new.so stable_ts_amd/libswx.so; fi
        ( timeout 900 "$@" ) >> gpurun_out/r06_c13_${tag}_${lib}.txt 2>> gpurun_out/r06_c13_${tag}.err
    done
    cp /tmp/libswx_new.so stable_ts_amd/libswx.so
}
run_pair kb_flash python scripts/kernel_bench.py --only flash --iters 100
run_pair kb_cross python scripts/kernel_bench.py --only cross --iters 200
run_pair bench python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline
run_pair align python bench.py --mode align --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline
for t in kb_flash kb_cross; do for l in bpermute new; do echo "== $t $l"; cat gpurun_out/r06_c13_${t}_${l}.txt; done; done
python - <<'PY'
import json
for t in ("bench", "align"):
    for l in ("bpermute", "new"):
        rows = [json.loads(x) for x in open(f"gpurun_out/r06_c13_{t}_{l}.txt") if x.startswith("{")]
        print(t, l, [r["ms_per_step"] for r in rows], [r["config"].get("words") for r in rows])
PY
