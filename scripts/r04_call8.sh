#!/bin/bash
# GPU call 8: fused cross-attention query projection (bit-identity tests + A/B), depth tests on two inputs
mkdir -p gpurun_out
echo "== fused xq tests"; timeout 600 python -m pytest tests/test_gpu_model.py -q --timeout=500 --tb=short -k "fused_cross_query or l2_prefetch or graph" 2>&1 | tail -6
echo "== A/B fused xq"; timeout 400 python scripts/ab_streams.py --flags 0,1048576 --rounds 4 --out gpurun_out/r04_c8_fused_xq_ab.json 2>&1 | tail -10
echo "== depth tests"; timeout 900 python -m pytest tests/test_gpu_f16_depth.py -q --timeout=800 --tb=short -rf 2>&1 | tail -15
cp gpurun_out/f16_depth_report.json gpurun_out/r04_f16_depth_report.json 2>/dev/null
