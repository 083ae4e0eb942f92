#!/bin/bash
# Round 5, GPU call 14: decode_beam_update_kernel with the candidates fetched and ranked in parallel: beam tests, then the headline
# and the sequential line (rocprofv3 summary of the latter for the kernel's new duration)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_largev3.py tests/test_gpu_batch_invariance.py tests/test_gpu_golden.py tests/test_gpu_f16_depth.py -m gpu -q -n 4 --timeout=400 --tb=short -rf 2>&1 | tail -8 ) | tee gpurun_out/r05_c14_beam_tests.log | cut -c1-250 | tail -4
( timeout 300 python bench.py --no-f32 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 ) | tee gpurun_out/r05_c14_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
bash scripts/rocprof_kernels.sh r05_c14_seq python $R/bench.py --sequential --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-f32
tail -1 gpurun_out/r05_c14_seq_cmd.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sequential', d['value'], d['ms_per_step'])"; grep "beam_update\|select_reg" gpurun_out/r05_c14_seq_kernels.csv
