#!/bin/bash
# Round 5, GPU call 5 = call 4 again after its fault: gemm_f32_rows64's idle tail re-loads landed in registers hipcc had already
# handed to the epilogue (fixed: every staging register is kept allocated up to the final wait; scripts/isa_asm_load_audit.py);
# gemm_f32_tiled is back on compiler-managed loads.  Every step in its own process, the f32 steps last.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
step() { echo "== $1 ($(date +%T))"; }
step "kernel tests: dec GEMM (ticket) first, then the rest"
( timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=280 --tb=short -k "dec_gemm" 2>&1 | tail -6 ) | tee gpurun_out/r05_c5_kernel_tests_dec.log | cut -c1-300 | tail -5
( timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=280 --tb=short -k "not dec_gemm" 2>&1 | tail -8 ) | tee gpurun_out/r05_c5_kernel_tests.log | cut -c1-300 | tail -6
step "bench f16 + ticket A/B (no f32 leg)"
( timeout 300 python bench.py --no-f32 --ab-flags 2097152 2>gpurun_out/r05_c5_bench.err | tail -1 ) > gpurun_out/r05_c5_bench_ticket_ab.json
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r05_c5_bench_ticket_ab.json'))
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'ab')}, d.get('roofline', {}).get('frac'))
except Exception as e:
    print("no bench line:", e); print(open('gpurun_out/r05_c5_bench.err').read()[-600:])
PY
step "model tests"
( timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_largev3.py tests/test_gpu_batch_invariance.py tests/test_gpu_golden.py -m gpu -q -n 4 --timeout=500 --tb=short -rf 2>&1 | tail -25 ) | tee gpurun_out/r05_c5_model_tests.log | cut -c1-300 | tail -8
step "strict f32 bench"
( timeout 200 python bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 2>gpurun_out/r05_c5_bench_f32.err | tail -1 ) | tee gpurun_out/r05_c5_bench_f32.json | cut -c1-250
step "f32 pass kernels"
timeout 200 bash scripts/rocprof_kernels.sh r05_c5_f32pass python $R/bench.py --dtype f32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
tail -1 gpurun_out/r05_c5_f32pass_cmd.log | cut -c1-200; head -20 gpurun_out/r05_c5_f32pass_kernels.csv
step done
