#!/bin/bash
# round 3, call 14: (a) the 256 x 256 two-stage GEMM: identity + repeat screen, micro-benchmark at large M, headline A/B
# (debug flag 131072 = never that kernel); (b) decode_select with the top-(G+1) from per-thread maxima: the decode parity tests
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c14
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
echo "host kernel $(uname -r)" | tee $O/box.txt
timeout 900 python tests/hw_checks/gemm_glds_check.py 2>&1 | grep -v amdgpu.ids > $O/gemm_big_check.txt; echo "gemm check rc=${PIPESTATUS[0]}" | tee -a $O/gemm_big_check.txt
grep -c "^ok" $O/gemm_big_check.txt; grep -v "^ok" $O/gemm_big_check.txt | head
timeout 300 python scripts/kernel_bench.py --only gemm_big 2>&1 | grep -v amdgpu.ids > $O/kb_gemm_big.txt; cat $O/kb_gemm_big.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -q -m gpu -k "decode or select or beam or greedy or gemm" -p no:cacheprovider > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for F in 131072 0 131072 0; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline --debug-flags $F > $O/bench_default_f${F}_$(date +%s).json 2>> $O/bench.err
done
for f in $O/bench_default_f*.json; do python -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j.get('debug_flags',0), j['ms_per_step'], j['value'])"; done
