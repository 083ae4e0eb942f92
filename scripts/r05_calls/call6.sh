#!/bin/bash
# Round 5, GPU call 6: the complete GPU suite on the final tree (in-launch slab reduction off by default, f32 kernels of this round,
# located near-tie criteria of tests/test_gpu_f16_bench_windows.py), complete log with its summary line, then smoke.
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/f16_bench_windows_report.json gpurun_out/f16_depth_report.json
( timeout 1100 python -m pytest tests -m gpu -q -n 4 --timeout=900 --tb=short -rf --durations=15 2>&1 | tail -60 ) | tee gpurun_out/r05_c6_gpu_suite.log | cut -c1-300 | tail -30
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) | tee gpurun_out/r05_c6_smoke.log
