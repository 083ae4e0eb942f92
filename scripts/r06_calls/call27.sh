#!/bin/bash
# round 6 call 27: gelu_erf2 (packed, branch-free GELU of the GEMM epilogues) -- exhaustive bit-identity over all 2^32 floats, kernel tests,
# then two builds alternating on one box (scripts/exp/libswx_gelu_scalar.so = -DSWX_GELU_SCALAR): the GELU launch at realistic magnitudes,
# headline pass, align()
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -4 ) > gpurun_out/r06_c27_tests.log; cat gpurun_out/r06_c27_tests.log
cp stable_ts_amd/libswx.so /tmp/libswx_new.so
for lib in gelu_scalar new gelu_scalar new; do
    if [ $lib = gelu_scalar ]; then cp scripts/exp/libswx_gelu_scalar.so stable_ts_amd/libswx.so; else cp /tmp/libswx_new.so stable_ts_amd/libswx.so; fi
    ( timeout 600 python scripts/kernel_bench.py --only gemm_gelu --iters 200 ) >> gpurun_out/r06_c27_kb_gelu_${lib}.txt 2>> gpurun_out/r06_c27.err
    ( timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline ) >> gpurun_out/r06_c27_bench_${lib}.txt 2>> gpurun_out/r06_c27.err
    ( timeout 900 python bench.py --mode align --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) >> gpurun_out/r06_c27_align_${lib}.txt 2>> gpurun_out/r06_c27.err
done
cp /tmp/libswx_new.so stable_ts_amd/libswx.so
for l in gelu_scalar new; do echo "== kb $l"; grep "std" gpurun_out/r06_c27_kb_gelu_${l}.txt; done
python - <<'PY'
import json
for t in ("bench", "align"):
    for l in ("gelu_scalar", "new"):
        rows = [json.loads(x) for x in open(f"gpurun_out/r06_c27_{t}_{l}.txt") if x.startswith("{")]
        print(t, l, [r["ms_per_step"] for r in rows], [r["config"].get("words") for r in rows])
PY
