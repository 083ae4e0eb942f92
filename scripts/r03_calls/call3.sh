#!/bin/bash
# round 3, call 3: head-selection kernels (f4) + the tests whose bars changed, graph statistics, the sequential mode's phase
# times, the host-contention probe, then the measurement set of the round (PMC traffic, rocprofv3 kernel summary, bench line)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c3
mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
stamp() { echo "=== $1 (t+$(( $(date +%s) - t0 ))s)"; }
stamp "head-selection kernels: hardware check"
timeout 600 python tests/hw_checks/score_qk_check.py > $O/score_qk_check.log 2>&1; echo "rc=$?"; tail -14 $O/score_qk_check.log
stamp "tests with new bars / new paths"
timeout 900 python -m pytest tests/test_gpu_f16_depth.py tests/test_gpu_largev3.py tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_kernels.py -q -m gpu \
    -k "words_vs_oracle or unsaturated or graph_replay or variants or new_kernel_paths or pending_device" -p no:cacheprovider > $O/pytest_sel.log 2>&1
tail -8 $O/pytest_sel.log
stamp "sequential mode: phase times + graph statistics"
timeout 300 python bench.py --sequential --steps 1 --warmup 1 --no-f32 --no-cpu-baseline --no-roofline --phase-times --host-profile $O/sequential_host_profile.txt > $O/bench_sequential.json 2> $O/bench_sequential.err
head -c 1500 $O/bench_sequential.json; echo
stamp "host contention probe (8 host-only replays at once)"
timeout 600 python scripts/host_contention_probe.py --procs 8 --passes 5 > $O/host_contention.json 2> $O/host_contention.err
cat $O/host_contention.json; tail -3 $O/host_contention.err
stamp "measurement set"
SKIP_SUITE=1 bash scripts/gpu_final.sh r03 > $O/gpu_final.log 2>&1
tail -5 $O/gpu_final.log | cut -c1-1500
stamp "done"
