#!/bin/bash
# round 6 call 18: price check of the decode step's vocabulary projection (tiled kernel vs the weight-streaming dec kernel at N = 51840),
# and the prefetch chain re-measured on today's tree (flag 32768 = no prefetch) on the headline pass and at 5 rows
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python scripts/kernel_bench.py --only logits --iters 100 ) > gpurun_out/r06_c18_kb_logits.txt 2> gpurun_out/r06_c18_kb.err
cat gpurun_out/r06_c18_kb_logits.txt; tail -3 gpurun_out/r06_c18_kb.err
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 32768 > gpurun_out/r06_c18_bench_prefetch_ab.json 2> gpurun_out/r06_c18_bench.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c18_bench_prefetch_ab.json'));print('headline prefetch A/B (on = NO prefetch)',d['value'],d['ms_per_step'],d.get('ab'))"
( timeout 600 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 32768 > gpurun_out/r06_c18_bench_seq_prefetch_ab.json 2> gpurun_out/r06_c18_seq.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c18_bench_seq_prefetch_ab.json'));print('sequential prefetch A/B (on = NO prefetch)',d['value'],d['ms_per_step'],d.get('ab'))"
