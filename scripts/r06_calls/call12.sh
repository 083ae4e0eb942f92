#!/bin/bash
# round 6 call 12: which cross-attention stream variant: 1 = two blocks + nt, 2 = two blocks, 3 = three blocks + nt (A/B each vs round 5's loop)
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "two_blocks_in_flight" 2>&1 | tail -3 )
for F in 33554432 67108864 100663296; do
  ( timeout 600 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags $F > gpurun_out/r06_c12_bench_xattn_ab_$F.json 2> gpurun_out/r06_c12_bench.err )
  python -c "
import json;d=json.load(open('gpurun_out/r06_c12_bench_xattn_ab_$F.json'));print('flags $F',d['ms_per_step'],d.get('ab'))"
done
