#!/bin/bash
# round 6 call 35: multi-token self-attention with 4-8 tokens of a (row, head) per workgroup and K / V staged in LDS once
# (self_attn_cached_mq_f16; flag 256 = SWX_FLAG_SELFATTN_NO_MQ = one wave per (row, token, head) from L2): kernel-level bit-identity, model
# tests incl. batch invariance and the bench-window parity file, A/B on the headline pass, at 120 windows and in align(); the scoring-pass
# cross-attention's four-group form now compiles for two workgroups per CU (call 34)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_batch_invariance.py tests/test_gpu_f16_bench_windows.py tests/test_gpu_golden.py -m gpu -q -x 2>&1 | tail -3 ) > gpurun_out/r06_c35_tests.log; cat gpurun_out/r06_c35_tests.log
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 256 > gpurun_out/r06_c35_bench_selfattn_mq_ab.json 2> gpurun_out/r06_c35.err )
( timeout 600 python bench.py --minutes 60 --batch 120 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 256 > gpurun_out/r06_c35_bench_b120_selfattn_mq_ab.json 2>> gpurun_out/r06_c35.err )
( timeout 600 python bench.py --mode align --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 256 > gpurun_out/r06_c35_bench_align_selfattn_mq_ab.json 2>> gpurun_out/r06_c35.err )
( timeout 900 bash scripts/rocprof_kernels.sh r06_c35_pass python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline ) > gpurun_out/r06_c35_rocprof.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_c35_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r06_c35_bench_")[1], d["value"], d["ms_per_step"], d["config"].get("words"), d.get("ab"))
    except Exception as e:
        print(f, "unreadable", e)
PY
grep "self_attn_cached\|attn_decode_cross2" gpurun_out/r06_c35_pass_kernels.csv | cut -c1-170
tail -3 gpurun_out/r06_c35.err
