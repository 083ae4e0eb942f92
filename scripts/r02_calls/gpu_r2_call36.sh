#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python bench.py 2>&1 | tail -1 ) | tee gpurun_out/bench_final_b.json | cut -c1-700
