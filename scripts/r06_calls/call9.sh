#!/bin/bash
# round 6 call 9: cross-K/V from one launch (EPI_KV) + the silence probe spread over the chip: bit-identity tests, align() A/B
# (flags 8388608 | 16777216 = the round-5 paths), headline A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "cross_kv or loudness" 2>&1 | tail -8 ) > gpurun_out/r06_c9_tests.log
cat gpurun_out/r06_c9_tests.log
( timeout 600 python bench.py --mode align --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 25165824 > gpurun_out/r06_c9_bench_align_ab.json 2> gpurun_out/r06_c9_align.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c9_bench_align_ab.json'));print('align',d['value'],d['ms_per_step'],d.get('ab'))"
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 25165824 > gpurun_out/r06_c9_bench_ab.json 2> gpurun_out/r06_c9_bench.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c9_bench_ab.json'));print('headline',d['value'],d['ms_per_step'],d.get('ab'))"
( timeout 600 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 25165824 > gpurun_out/r06_c9_bench_seq_ab.json 2> gpurun_out/r06_c9_seq.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c9_bench_seq_ab.json'));print('sequential',d['value'],d['ms_per_step'],d.get('ab'))"
