#!/bin/bash
# round 6 call 2: grouped tile order of gemm_f16_big8 -- bit-identity on hardware in both orders, micro-benchmark A/B, pass A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python tests/hw_checks/gemm_big8_check.py --reps 10 2>&1 | tail -25 ) > gpurun_out/r06_c2_big8_check.txt
( timeout 600 python scripts/kernel_bench.py --only gemm_big 2>&1 | tail -20 ) > gpurun_out/r06_c2_kb_gemm_big.txt
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 4194304 > gpurun_out/r06_c2_bench_big8_order_ab.json 2> gpurun_out/r06_c2_bench.err )
cat gpurun_out/r06_c2_big8_check.txt gpurun_out/r06_c2_kb_gemm_big.txt
python -c "
import json;d=json.load(open('gpurun_out/r06_c2_bench_big8_order_ab.json'));print(d['ms_per_step'],d.get('ab'))"
