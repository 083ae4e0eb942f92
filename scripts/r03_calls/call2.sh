#!/bin/bash
# round 3, call 2: the whole GPU suite on the pruned library + the captured decode-step graph, then graph A/B lines
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c2
mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
stamp() { echo "=== $1 (t+$(( $(date +%s) - t0 ))s)"; }
stamp "graph bit-identity first (fast fail)"
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "graph_replay" -p no:cacheprovider > $O/pytest_graph.log 2>&1
tail -5 $O/pytest_graph.log
stamp "sequential / spans with and without the graph"
for F in 0 16384; do
  timeout 300 python bench.py --sequential --steps 1 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline --debug-flags $F > $O/bench_sequential_f$F.json 2> $O/bench_sequential_f$F.err
  head -c 260 $O/bench_sequential_f$F.json; echo
  timeout 300 python bench.py --spans 20 --steps 2 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline --debug-flags $F > $O/bench_spans20_f$F.json 2> $O/bench_spans20_f$F.err
  head -c 260 $O/bench_spans20_f$F.json; echo
  timeout 300 python bench.py --steps 5 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline --debug-flags $F > $O/bench_default_f$F.json 2> $O/bench_default_f$F.err
  head -c 260 $O/bench_default_f$F.json; echo
done
stamp "base.en single window, graph"
timeout 300 python bench.py --model base.en --minutes 0.5 --batch 1 --beam 1 --steps 20 --warmup 3 --no-f32 --no-cpu-baseline --no-roofline > $O/bench_base_en_1win.json 2> $O/bench_base_en_1win.err
head -c 260 $O/bench_base_en_1win.json; echo
stamp "full GPU suite"
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log
stamp "smoke"
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
stamp "done"
