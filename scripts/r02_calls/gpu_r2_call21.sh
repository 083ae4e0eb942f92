#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python scripts/micro/align_trace.py 2>&1 | tail -12 | tee gpurun_out/align_trace.txt
