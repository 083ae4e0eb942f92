#!/bin/bash
# round 6 call 30: (a) gemm_dec_f16<.., WPB = 1> -- decode-step GEMMs of launches with <= 80 workgroups as single-wave workgroups (flag 16 =
# SWX_FLAG_DEC_NO_W1 puts the four-wave workgroups back): bit-identity per epilogue, A/B in the sequential mode; (b) experiment: the
# many-window cross-attention with its second key block requested inside the query projection (flag 8 = SWX_FLAG_XATTN_EARLY2): A/B on
# the headline pass; (c) kernel table of the sequential mode on this tree
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_batch_invariance.py -m gpu -q -x 2>&1 | tail -5 ) > gpurun_out/r06_c30_tests.log; cat gpurun_out/r06_c30_tests.log
( timeout 900 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 16 > gpurun_out/r06_c30_bench_seq_dec_w1_ab.json 2> gpurun_out/r06_c30.err )
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 8 > gpurun_out/r06_c30_bench_xattn_early2_ab.json 2>> gpurun_out/r06_c30.err )
( timeout 600 python bench.py --minutes 60 --batch 120 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 8 > gpurun_out/r06_c30_bench_b120_xattn_early2_ab.json 2>> gpurun_out/r06_c30.err )
( timeout 600 python bench.py --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 16 > gpurun_out/r06_c30_bench_base_en_dec_w1_ab.json 2>> gpurun_out/r06_c30.err )
( timeout 900 bash scripts/rocprof_kernels.sh r06_c30_seq python bench.py --sequential --minutes 2 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) > gpurun_out/r06_c30_rocprof.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c30_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c30_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
head -14 gpurun_out/r06_c30_seq_kernels.csv | cut -c1-170
tail -3 gpurun_out/r06_c30.err
