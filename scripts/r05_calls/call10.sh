#!/bin/bash
# Round 5, GPU call 10: the child-process hardware checks with the thread pool capped at the CPU quota (they ran 256 OpenMP workers
# under a 16-CPU quota: 145 + 65 + 44 + 45 + 19 s of the 545-s serial suite)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -q -m gpu -k "subprocess or seam_b3" --durations=8 2>&1 | tail -16 ) | tee gpurun_out/r05_c10_subprocess_checks.log | cut -c1-200
