#!/bin/bash
# round 6 call 15: (1) the decode-chain seam experiment of VERDICT r5 item 4 (scripts/micro/dec_pair_fused.hip: two chained dec GEMMs as
# one persistent launch with a run-ahead weight loader vs two dependent launches), wrapped in its own timeout; (2) beam bookkeeping
# without the commit launch + one-wave step finish: decode tests; (3) headline bench + rocprofv3 kernel summary of the current tree
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
( timeout 120 scripts/micro/dec_pair 64 ) > gpurun_out/r06_c15_dec_pair_fused.txt 2>&1
echo "dec_pair rc=$?" >> gpurun_out/r06_c15_dec_pair_fused.txt
cat gpurun_out/r06_c15_dec_pair_fused.txt
( timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_batch_invariance.py -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r06_c15_tests.log
cat gpurun_out/r06_c15_tests.log
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline > gpurun_out/r06_c15_bench.json 2> gpurun_out/r06_c15_bench.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c15_bench.json'));print('headline',d['value'],d['ms_per_step'],d['config']['words'])"
bash scripts/rocprof_kernels.sh r06_c15_pass python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline
head -40 gpurun_out/r06_c15_pass_kernels.csv | cut -c1-170
