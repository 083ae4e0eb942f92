#!/bin/bash
# round 6 call 4: decode-step self-attention -- beams of a window on one XCD + position / ancestor ids in one round trip: bit-identity
# on hardware, pass A/B (flag 8388608 = the round-5 launch)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python tests/hw_checks/self_attn_step_check.py 2>&1 | tail -16 ) > gpurun_out/r06_c4_self_attn_check.txt
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 8388608 > gpurun_out/r06_c4_bench_selfattn_ab.json 2> gpurun_out/r06_c4_bench.err )
cat gpurun_out/r06_c4_self_attn_check.txt
python -c "
import json;d=json.load(open('gpurun_out/r06_c4_bench_selfattn_ab.json'));print(d['ms_per_step'],d.get('ab'))"
tail -3 gpurun_out/r06_c4_bench.err
