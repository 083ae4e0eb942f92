#!/bin/bash
# round 6 call 14: attn_flash3_f16 (software-pipelined flash tile) -- bit-identity vs generation 2, kernel micro-benchmark and the
# headline / align() A/B inside one process each (flag 67108864 = generation 2); rocprofv3 kernel summary of the headline pass
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "flash3 or attention" 2>&1 | tail -8 ) > gpurun_out/r06_c14_tests.log
cat gpurun_out/r06_c14_tests.log
for i in 1 2; do
  ( timeout 600 python scripts/kernel_bench.py --only flash --iters 100 --flags 67108864 ) >> gpurun_out/r06_c14_kb_flash_gen2.txt 2>> gpurun_out/r06_c14_kb.err
  ( timeout 600 python scripts/kernel_bench.py --only flash --iters 100 ) >> gpurun_out/r06_c14_kb_flash_gen3.txt 2>> gpurun_out/r06_c14_kb.err
  ( timeout 600 python scripts/kernel_bench.py --only flash_small --iters 200 --flags 67108864 ) >> gpurun_out/r06_c14_kb_flash_small_gen2.txt 2>> gpurun_out/r06_c14_kb.err
  ( timeout 600 python scripts/kernel_bench.py --only flash_small --iters 200 ) >> gpurun_out/r06_c14_kb_flash_small_gen3.txt 2>> gpurun_out/r06_c14_kb.err
done
for t in flash flash_small; do for l in gen2 gen3; do echo "== $t $l"; cat gpurun_out/r06_c14_kb_${t}_${l}.txt; done; done
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 67108864 > gpurun_out/r06_c14_bench_flash_ab.json 2> gpurun_out/r06_c14_bench.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c14_bench_flash_ab.json'));print('headline',d['value'],d['ms_per_step'],d.get('ab'))"
( timeout 600 python bench.py --mode align --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 67108864 > gpurun_out/r06_c14_bench_align_flash_ab.json 2> gpurun_out/r06_c14_align.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c14_bench_align_flash_ab.json'));print('align',d['value'],d['ms_per_step'],d.get('ab'))"
bash scripts/rocprof_kernels.sh r06_c14_pass python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline
head -24 gpurun_out/r06_c14_pass_kernels.csv | cut -c1-170
