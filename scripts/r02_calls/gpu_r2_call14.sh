#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== align mode host profile"; ( timeout 600 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --host-profile gpurun_out/align_host_profile.txt 2>&1 | tail -1 | cut -c1-300 )
head -60 gpurun_out/align_host_profile.txt | cut -c1-160
echo "== transcribe host profile"; ( timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --host-profile gpurun_out/transcribe_host_profile.txt 2>&1 | tail -1 | cut -c1-300 )
head -50 gpurun_out/transcribe_host_profile.txt | cut -c1-160
