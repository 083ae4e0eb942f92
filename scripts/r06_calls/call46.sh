#!/bin/bash
# usage: gpurun --timeout 3000 -- 'bash scripts/r06_calls/call46.sh'
# x1 on EVERY window of the timed recording: tests/test_gpu_f16_bench_windows.py over windows 0 .. 19 (the suite keeps 0 / 7 / 19),
# f16 and strict f32, greedy + beam 5 + words; the per-case numbers land in gpurun_out/f16_bench_windows_report.json
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/f16_bench_windows_report.json
( SWX_BENCH_WINDOWS=all timeout 2800 python -m pytest tests/test_gpu_f16_bench_windows.py -q -m gpu --durations=8 -p no:cacheprovider 2>&1 | tail -60 ) > gpurun_out/r06_c46_all_windows.log
cp gpurun_out/f16_bench_windows_report.json gpurun_out/r06_c46_bench_windows_all20_report.json
tail -15 gpurun_out/r06_c46_all_windows.log
