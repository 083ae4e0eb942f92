#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 60 python __graft_entry__.py smoke 2>&1 | tail -3 ) > gpurun_out/smoke.log
( timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -n 4 --timeout=280 2>&1 | tail -15 ) > gpurun_out/model.log
( timeout 240 python bench.py 2> gpurun_out/bench.err | tail -2 ) > gpurun_out/bench.log
cat gpurun_out/smoke.log; tail -8 gpurun_out/model.log; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.log | cut -c1-2500
