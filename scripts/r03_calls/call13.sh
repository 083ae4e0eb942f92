#!/bin/bash
# round 3, call 13: the ring GEMM with the kept dispatch (depth 3; ring only up to one workgroup per CU): identity, micro-benchmark,
# align() and base.en lines; one-window encoder timing (device vs wall vs host idling in front)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c13
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
echo "host kernel $(uname -r)" | tee $O/box.txt
timeout 600 python tests/hw_checks/gemm_glds_check.py 2>&1 | grep -v amdgpu.ids > $O/gemm_ring_check.txt; echo "gemm check rc=${PIPESTATUS[0]}" | tee -a $O/gemm_ring_check.txt
tail -4 $O/gemm_ring_check.txt
timeout 300 python scripts/kernel_bench.py --only gemm_small 2>&1 | grep -v amdgpu.ids > $O/kb_gemm_small.txt; cat $O/kb_gemm_small.txt
timeout 300 python scripts/exp/encode_b1_timing.py 2>&1 | grep -v amdgpu.ids > $O/encode_b1_timing.txt; cat $O/encode_b1_timing.txt
for i in 1 2; do
  timeout 400 python bench.py --mode align --steps 3 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline --phase-times > $O/bench_align_$i.json 2>> $O/bench.err
done
timeout 400 python bench.py --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 20 --warmup 3 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_base_en_1win.json 2>> $O/bench.err
timeout 600 python bench.py --sequential --steps 2 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_sequential.json 2>> $O/bench.err
for f in $O/bench_*.json; do python -c "
import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], j['ms_per_step'], j['value'], json.dumps(j.get('phase_ms'))[:600])"; done
tail -3 $O/bench.err
