#!/bin/bash
# GPU call 14: LDS fragment reads eight steps ahead in the single-row-tile MFMA loops (tall GEMM, fused query projection, MT = 1 decode GEMMs)
mkdir -p gpurun_out
echo "== dec tall check"; timeout 300 python tests/hw_checks/dec_tall_check.py 2>&1 | tail -3
echo "== decode tests"; timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_batch_invariance.py -q --timeout=500 --tb=short 2>&1 | tail -4
echo "== A/B"; timeout 400 python scripts/ab_streams.py --flags 0,262144,1048576 --rounds 3 --phase --out gpurun_out/r04_c14_ab.json 2>&1 | tail -12
