#!/bin/bash
# Round 5, GPU call 12: second form of the in-launch slab reduction (write-through `sc1` slab stores + drained ticket + `sc1` reads, no
# L2 write-back / invalidate): bit-identity test, then A/B against the separate finish launch inside one process
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=180 --tb=short -k "dec_gemm" 2>&1 | tail -6 ) | tee gpurun_out/r05_c12_dec_tests.log | cut -c1-300 | tail -4
( timeout 300 python bench.py --no-f32 --no-cpu-baseline --no-roofline --steps 3 --warmup 1 --ab-flags 2097152 2>gpurun_out/r05_c12_bench.err | tail -1 ) > gpurun_out/r05_c12_bench_ticket_sc1_ab.json
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r05_c12_bench_ticket_sc1_ab.json'))
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'ab')})
except Exception as e:
    print("no bench line:", e); print(open('gpurun_out/r05_c12_bench.err').read()[-600:])
PY
