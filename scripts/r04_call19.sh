#!/bin/bash
# GPU call 19: K / V^T cache prefetch under the fused query projection: 0 / 2 / 4 blocks per wave
mkdir -p gpurun_out
echo "== decode tests"; timeout 600 python -m pytest tests/test_gpu_model.py -q --timeout=500 --tb=short -k "fused_cross_query or graph_replay or l2_prefetch" 2>&1 | tail -3
echo "== A/B"; timeout 400 python scripts/ab_streams.py --flags 0,2097152,4194304 --rounds 4 --out gpurun_out/r04_c19_kv_ahead_ab.json 2>&1 | tail -14
