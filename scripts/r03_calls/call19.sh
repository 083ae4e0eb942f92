#!/bin/bash
# round 3, call 19: after the dispatch refactoring (swx_gemm_plan_f16): the tiled-GEMM identity check, the kernel / model tests that
# go through swx_gemm, smoke, and the default bench line
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c19
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
echo "host kernel $(uname -r)" | tee $O/box.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -2 $O/pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 )
timeout 600 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline > $O/bench_default.json 2>> $O/bench.err
python -c "
import json; j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline']['frac'], j['kernel_time_ms'])"
