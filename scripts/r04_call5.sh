#!/bin/bash
# GPU call 5 of round 4: tall dec GEMM bit-identity, batch invariance, full-depth fp16 parity on the benchmark's recipe, real speech,
# A/B of the multi-token pass dispatch (0 = invariant: dec / tall GEMMs + grouped cross-attention, 262144 = round 3, 524288 = no tall kernel)
mkdir -p gpurun_out
echo "== dec tall check"; timeout 300 python tests/hw_checks/dec_tall_check.py 2>&1 | tail -16
echo "== parity tests"; timeout 900 python -m pytest tests/test_gpu_batch_invariance.py tests/test_gpu_f16_depth.py "tests/test_gpu_golden.py::test_real_speech_flac_matches_reference_glue" -q --timeout=800 --tb=short -rf 2>&1 | tail -25
cp gpurun_out/f16_depth_report.json gpurun_out/r04_f16_depth_report.json 2>/dev/null
echo "== dispatch A/B"; timeout 400 python scripts/ab_streams.py --flags 0,262144,524288 --rounds 3 --phase --out gpurun_out/r04_c5_score_dispatch_ab.json 2>&1 | tail -12
