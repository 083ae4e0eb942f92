#!/bin/bash
# usage: gpurun --timeout 1800 -- 'bash scripts/gpu_final_short.sh [tag]'   (gpu_final.sh without the two PMC passes)
tag=${1:-final}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
scripts/rocprof_kernels.sh bench_$tag python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32
head -16 gpurun_out/bench_${tag}_kernels.csv | cut -c1-150
echo "== smoke"; ( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 )
echo "== default bench line"; ( timeout 900 python bench.py 2>&1 | tail -1 ) | tee gpurun_out/bench_$tag.json | cut -c1-600
echo "== align line"; ( timeout 600 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 2>&1 | tail -1 ) | tee gpurun_out/bench_${tag}_align.json | cut -c1-330
echo "== gpu suite"; ( timeout 1500 python -m pytest tests -m gpu -q -n 4 --timeout=900 --tb=short -rf 2>&1 | tail -60 ) | tee gpurun_out/gpu_suite_$tag.log | tail -12
