#!/bin/bash
# round 6 call 17: the whole GPU suite on the current tree (serial: the driver's command), then A/Bs inside one process each:
# five-rows-per-workgroup self-attention (flag 134217728) on the headline pass, V transposed by the QKV epilogue
# (flag 268435456 = the separate launch) on align() and the sequential mode
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r06_c17_gpu_suite.log
cat gpurun_out/r06_c17_gpu_suite.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > gpurun_out/r06_c17_smoke.log; cat gpurun_out/r06_c17_smoke.log
( timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 134217728 > gpurun_out/r06_c17_bench_selfattn_wg5_ab.json 2> gpurun_out/r06_c17_bench.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c17_bench_selfattn_wg5_ab.json'));print('headline wg5 A/B (on = five rows per workgroup)',d['value'],d['ms_per_step'],d.get('ab'))"
( timeout 600 python bench.py --mode align --steps 2 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 268435456 > gpurun_out/r06_c17_bench_align_vt_ab.json 2> gpurun_out/r06_c17_align.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c17_bench_align_vt_ab.json'));print('align VT A/B (on = separate transpose launch)',d['value'],d['ms_per_step'],d.get('ab'))"
( timeout 600 python bench.py --sequential --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline --ab-flags 134217728 > gpurun_out/r06_c17_bench_seq_wg5_ab.json 2> gpurun_out/r06_c17_seq.err )
python -c "
import json;d=json.load(open('gpurun_out/r06_c17_bench_seq_wg5_ab.json'));print('sequential wg5 A/B',d['value'],d['ms_per_step'],d.get('ab'))"
