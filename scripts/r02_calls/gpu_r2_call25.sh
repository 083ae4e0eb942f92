#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 python scripts/micro/align_trace.py 2>&1 | tail -10 | cut -c1-700 ) | tee gpurun_out/align_trace3.txt
