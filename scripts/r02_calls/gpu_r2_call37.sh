#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 290 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/gpu_suite_sequential.log
