#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== glds hw check"; timeout 300 python tests/hw_checks/gemm_glds_check.py 2>&1 | tail -11
for k in 9 7; do echo "== gemm kernel $k"; timeout 200 python scripts/kernel_bench.py --only gemm --gemm-kernel $k 2>&1 | grep "M=  *\(1500\|2240\)" ; done | tee gpurun_out/kb_gemm_narrow.txt
echo "== model tests"; ( timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py -m gpu -q -n 4 --timeout=600 2>&1 | tail -3 )
echo "== align"; ( timeout 600 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
echo "== bench"; ( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
