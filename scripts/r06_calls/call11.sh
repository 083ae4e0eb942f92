#!/bin/bash
# round 6 call 11: decode-step cross-attention with two key blocks in flight per wave + nt loads (flag 33554432): bit-identity, A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "two_blocks_in_flight or fused_cross_query" 2>&1 | tail -5 ) > gpurun_out/r06_c11_tests.log
cat gpurun_out/r06_c11_tests.log
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 33554432 > gpurun_out/r06_c11_bench_xattn_ab.json 2> gpurun_out/r06_c11_bench.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c11_bench_xattn_ab.json'));print('headline',d['value'],d['ms_per_step'],d.get('ab'))"
( timeout 600 python bench.py --minutes 60 --batch 120 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 33554432 > gpurun_out/r06_c11_bench_b120_xattn_ab.json 2> gpurun_out/r06_c11_b120.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c11_bench_b120_xattn_ab.json'));print('b120',d['value'],d['ms_per_step'],d.get('ab'))"
