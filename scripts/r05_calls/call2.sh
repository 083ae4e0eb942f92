#!/bin/bash
# Round 5, GPU call 2: the new x1 parity file (windows 0 / 7 / 19 of the timed recording vs the f32 oracle) -- first hardware run
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/f16_bench_windows_report.json
( timeout 900 python -m pytest tests/test_gpu_f16_bench_windows.py -m gpu -q --timeout=880 --tb=short -rf --durations=12 2>&1 | tail -60 ) | tee gpurun_out/r05_c2_bench_windows.log | cut -c1-400 | tail -45
