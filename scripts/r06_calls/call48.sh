#!/bin/bash
# usage: gpurun --timeout 1500 -- 'bash scripts/r06_calls/call48.sh'
# The model families that had never run on hardware: small.en (d = 768), medium (d = 1024), large-v3-turbo (4 decoder layers) --
# tests/test_gpu_model_families.py (strict f32 vs the oracle, fp16 step kernels, fp16 batch invariance)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1300 python -m pytest tests/test_gpu_model_families.py -q -m gpu --durations=20 -p no:cacheprovider 2>&1 | tail -70 ) > gpurun_out/r06_c48_model_families.log
tail -40 gpurun_out/r06_c48_model_families.log
