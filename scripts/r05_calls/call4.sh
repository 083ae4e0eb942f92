#!/bin/bash
# Round 5, GPU call 4: (1) kernel tests (f32 GEMMs on asm loads + counted waits; in-launch slab reduction = DEC_TICKET, first hardware
# run), (2) the strict-f32 model tests on the new GEMM pipelines, (3) default bench line with the ticket A/B inside one process and
# the strict-f32 leg, (4) prefetch-chain A/B at batch 120 (never A/B'd where the step is bandwidth-bound), (5) f32 profile
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
step() { echo "== $1 ($(date +%T))"; }
step "kernel tests"
( timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=380 --tb=short 2>&1 | tail -15 ) | tee gpurun_out/r05_c4_kernel_tests.log | cut -c1-300 | tail -12
step "strict f32 + decode tests"
( timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_largev3.py tests/test_gpu_batch_invariance.py tests/test_gpu_golden.py -m gpu -q -n 4 --timeout=500 --tb=short -rf 2>&1 | tail -25 ) | tee gpurun_out/r05_c4_model_tests.log | cut -c1-300 | tail -12
step "bench + ticket A/B"
( timeout 400 python bench.py --ab-flags 2097152 2>gpurun_out/r05_c4_bench.err | tail -1 ) | tee gpurun_out/r05_c4_bench_ticket_ab.json | cut -c1-300
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_c4_bench_ticket_ab.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'ab', 'strict_f32')})
PY
step "prefetch A/B at batch 120"
( timeout 400 python bench.py --minutes 60 --batch 120 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 --ab-flags 32768 2>/dev/null | tail -1 ) | tee gpurun_out/r05_c4_bench_b120_prefetch_ab.json | cut -c1-200
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_c4_bench_b120_prefetch_ab.json'))
print({k: d.get(k) for k in ('value', 'ms_per_step', 'ab')})
PY
step "f32 pass kernels"
bash scripts/rocprof_kernels.sh r05_c4_f32pass python $R/bench.py --dtype f32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
tail -1 gpurun_out/r05_c4_f32pass_cmd.log | cut -c1-300; head -24 gpurun_out/r05_c4_f32pass_kernels.csv
step done
