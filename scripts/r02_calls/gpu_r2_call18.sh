#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --phase-times 2>&1 | tail -1 ) | tee gpurun_out/bench18_align.json | cut -c1-1700
