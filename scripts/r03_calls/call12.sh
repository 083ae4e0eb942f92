#!/bin/bash
# round 3, call 12: (a) XCD-hierarchical grid barrier vs the flat counter barrier vs a dependent launch;
# (b) the ring GEMM for launches with few tiles (encoder at batch 1): bit-identity + repeat screen, micro-benchmark by kernel
# code, align() A/B (debug flag 65536 = never the ring kernel) with stage times, per-kernel trace of the align bench;
# (c) flash attention at 1 / 2 / 4 windows by queries per wave
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c12
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
echo "host kernel $(uname -r)" | tee $O/box.txt
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/grid_barrier_bench.hip -o /tmp/gbar && timeout 120 /tmp/gbar ) > $O/grid_barrier_xcd.txt 2>&1
cat $O/grid_barrier_xcd.txt
timeout 600 python tests/hw_checks/gemm_glds_check.py > $O/gemm_ring_check.txt 2>&1; echo "gemm check rc=$?" | tee -a $O/gemm_ring_check.txt
cat $O/gemm_ring_check.txt
timeout 300 python scripts/kernel_bench.py --only gemm_small > $O/kb_gemm_small.txt 2>&1; cat $O/kb_gemm_small.txt
timeout 300 python scripts/kernel_bench.py --only flash_small > $O/kb_flash_small.txt 2>&1; cat $O/kb_flash_small.txt
for F in 65536 0 65536 0; do
  timeout 400 python bench.py --mode align --steps 3 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline --phase-times --debug-flags $F > $O/bench_align_f${F}_$(date +%s).json 2>> $O/bench.err
done
for f in $O/bench_align_f*.json; do python -c "
import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j.get('debug_flags',0), j['ms_per_step'], j['value'], json.dumps(j.get('phase_ms'))[:700])"; done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -q -m gpu -k "gemm or flash or attention or align or encoder" -p no:cacheprovider > $O/pytest.log 2>&1
tail -3 $O/pytest.log
scripts/rocprof_kernels.sh align_c12 python $R/bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-f32 > $O/rocprof.log 2>&1
head -25 gpurun_out/align_c12_kernels.csv
