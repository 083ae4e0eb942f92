#!/bin/bash
# round 3, call 11: the prefetch with the linear (placement-agnostic) deal: identity tests + A/B
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c11
mkdir -p $O
export PYTHONUNBUFFERED=1
echo "host kernel $(uname -r)"
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -q -m gpu -k "prefetch or graph_replay or dec_gemm or dec_step or small_pass" -p no:cacheprovider > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for F in 0 32768 0 32768; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline --debug-flags $F > $O/bench_default_f${F}_$(date +%s).json 2>> $O/bench.err
done
for f in $O/bench_default_f*.json; do python -c "import json,sys; j=json.load(open('$f')); print(j.get('debug_flags',0), j['ms_per_step'], j['value'])"; done
