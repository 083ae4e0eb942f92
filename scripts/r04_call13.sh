#!/bin/bash
# GPU call 13: three-buffer tall GEMM (bit-identity x16) + self-attention step check (fixed ancestor table) + dispatch A/B
mkdir -p gpurun_out
echo "== dec tall check"; timeout 300 python tests/hw_checks/dec_tall_check.py 2>&1 | tail -13
echo "== self-attn step check"; timeout 200 python tests/hw_checks/self_attn_step_check.py 2>&1 | tail -12
echo "== invariance"; timeout 600 python -m pytest tests/test_gpu_batch_invariance.py -q --timeout=500 --tb=short 2>&1 | tail -3
echo "== dispatch A/B"; timeout 400 python scripts/ab_streams.py --flags 0,262144 --rounds 3 --phase --out gpurun_out/r04_c13_score_dispatch_ab.json 2>&1 | tail -9
