#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
thr() { grep -E "nr_throttled" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo; }
thr
echo "== align"; ( timeout 600 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 2>&1 | tail -1 ) | tee gpurun_out/bench29_align.json | cut -c1-330; thr
echo "== spans"; ( timeout 600 python bench.py --spans 20 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 ); thr
echo "== sequential"; ( timeout 400 python bench.py --sequential --minutes 3 --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 ); thr
echo "== bench"; ( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 ); thr
echo "== golden tests"; ( timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -q -n 4 --timeout=600 2>&1 | tail -3 )
