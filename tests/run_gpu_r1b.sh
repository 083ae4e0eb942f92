#!/bin/bash
# round-1 follow-up: validate the decode-step switches (bit-identity tests) and time them on the bench workload
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 240 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -n 4 --timeout=200 -k "switches or fast_step or gemm or attention or attn" 2>&1 | tail -12 ) > gpurun_out/switch_tests.log
( timeout 150 python tests/tune_flags.py --flags 0,20,32,52,0,52 2>&1 | grep -v "^\[" | tail -12 ) > gpurun_out/tune_flags.log
tail -12 gpurun_out/switch_tests.log; cat gpurun_out/tune_flags.log
