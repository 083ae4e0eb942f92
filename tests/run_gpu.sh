#!/bin/bash
# helper for gpurun: runs the GPU test tiers and keeps the logs under gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -m gpu -q -n 2 --timeout=600 2>&1 | tail -40 > gpurun_out/kernels.log
echo "kernels exit: ${PIPESTATUS[0]}" >> gpurun_out/kernels.log
python -m pytest tests/test_gpu_model.py -m gpu -q -n 2 --timeout=900 2>&1 | tail -80 > gpurun_out/model.log
echo "model exit: ${PIPESTATUS[0]}" >> gpurun_out/model.log
tail -5 gpurun_out/kernels.log; tail -30 gpurun_out/model.log
