#!/bin/bash
# round 3, call 10: HIP_FORCE_DEV_KERNARG (kernel arguments in device memory) A/B, graph replay and eager
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c10
mkdir -p $O
export PYTHONUNBUFFERED=1
for K in 0 1 0 1; do
  HIP_FORCE_DEV_KERNARG=$K timeout 300 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_graph_k${K}_$(date +%s).json 2>> $O/bench.err
  HIP_FORCE_DEV_KERNARG=$K timeout 300 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline --debug-flags 16384 > $O/bench_eager_k${K}_$(date +%s).json 2>> $O/bench.err
done
HIP_FORCE_DEV_KERNARG=1 timeout 300 python bench.py --sequential --steps 1 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_sequential_k1.json 2>> $O/bench.err
timeout 300 python bench.py --sequential --steps 1 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_sequential_kdef.json 2>> $O/bench.err
for f in $O/bench_*.json; do python -c "import json,sys; j=json.load(open('$f')); print('$f'.split('/')[-1], j['ms_per_step'], j['value'])"; done
