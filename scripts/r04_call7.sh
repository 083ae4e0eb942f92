#!/bin/bash
# GPU call 7: tall kernel with the relaxed wait (bit-identity x16 reps), full-depth fp16 tests on the benchmark's audio recipe,
# dispatch A/B, host tape for offline profiling of the host tail
mkdir -p gpurun_out
echo "== dec tall check"; timeout 300 python tests/hw_checks/dec_tall_check.py 2>&1 | tail -14
echo "== depth tests"; timeout 900 python -m pytest tests/test_gpu_f16_depth.py -q --timeout=800 --tb=short -rf 2>&1 | tail -15
cp gpurun_out/f16_depth_report.json gpurun_out/r04_f16_depth_report.json 2>/dev/null
echo "== dispatch A/B"; timeout 400 python scripts/ab_streams.py --flags 0,262144 --rounds 3 --phase --out gpurun_out/r04_c7_score_dispatch_ab.json 2>&1 | tail -9
echo "== host tape"; timeout 300 python scripts/host_contention_probe.py --record-only --tape gpurun_out/r04_host_tape.pkl 2>&1 | tail -2; ls -la gpurun_out/r04_host_tape.pkl
