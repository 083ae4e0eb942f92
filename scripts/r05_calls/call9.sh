#!/bin/bash
# Round 5, GPU call 9: the driver's own command for the GPU tier, serial (no xdist), on the final tree -- how long it takes and the
# complete log with its summary line
mkdir -p gpurun_out
export TMPDIR=/tmp
SECONDS=0
( timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 2>&1 | tail -40 ) | tee gpurun_out/r05_c9_gpu_suite_serial.log | cut -c1-250 | tail -22
echo "serial suite wall: ${SECONDS}s" | tee -a gpurun_out/r05_c9_gpu_suite_serial.log
