#!/bin/bash
# usage: gpurun --timeout 1800 -- 'bash scripts/gpu_experiments.sh'
# First hardware run of everything that was written without GPU access (end of round 1): correctness checks first, then
# the A/B timings that decide which experiments become defaults.  Every step writes its own log under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout ${T:-240} "$@" > /tmp/swx_step.log 2>&1; rc=$?; tail -${L:-25} /tmp/swx_step.log > gpurun_out/$name.log; echo "== $name: exit $rc"; cat gpurun_out/$name.log; }

# 1. correctness of the new device paths (each exits 0 on parity)
run chk_splitk   python tests/hw_checks/splitk_hook_check.py
run chk_glds     python tests/hw_checks/gemm_glds_check.py
run chk_melrag   python tests/hw_checks/mel_ragged_check.py
run chk_scoreqk  python tests/hw_checks/score_qk_check.py
run chk_b3       python tests/hw_checks/b3_check.py
SWX_PG_POLICY="1536x384=1,1152x384=1,384x384=1,384x1536=2" run chk_policy_tiny python tests/hw_checks/pg_policy_check.py tiny.en
SWX_PG_POLICY="2048x512=1,1536x512=1,512x512=2,512x2048=4" run chk_policy_base python tests/hw_checks/pg_policy_check.py base.en

# 2. per-kernel timings: register-staged vs direct-to-LDS tiled GEMM; decode GEMM + finish under K-split policies
run kb_gemm1 python scripts/kernel_bench.py --only gemm --gemm-kernel 1
run kb_gemm4 python scripts/kernel_bench.py --only gemm --gemm-kernel 4
run kb_splitk_default python scripts/kernel_bench.py --only splitk
SWX_PG_POLICY="5120x1280=1,3840x1280=1" run kb_splitk_fat_mlp1_qkv python scripts/kernel_bench.py --only splitk
SWX_PG_POLICY="5120x1280=1,3840x1280=1,1280x1280=5,1280x5120=8" run kb_splitk_fat_all python scripts/kernel_bench.py --only splitk
SWX_PG_POLICY="5120x1280=2,3840x1280=2,1280x1280=4,1280x5120=4" run kb_splitk_half python scripts/kernel_bench.py --only splitk
run kb_attn python scripts/kernel_bench.py --only cross

# 3. whole-pass A/B on the bench workload (one process per environment setting; flags switch in-process)
T=200 run pass_flags python scripts/tune_flags.py --flags 84,340,84,340
SWX_PG_POLICY="5120x1280=1,3840x1280=1" T=200 run pass_fat python scripts/tune_flags.py --flags 84,340
SWX_PG_POLICY="5120x1280=1,3840x1280=1,1280x1280=5,1280x5120=8" T=200 run pass_fat_all python scripts/tune_flags.py --flags 84,340
L=3 T=200 run bench_streams2 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --streams 2
L=3 T=200 run bench_rich python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --embed-gain 3 --ts-gain 0.01 --max-instant-words 1
# span-parallel mode (exact per-span sequential semantics, spans.py): first device run + its rate next to the window batches
L=3 T=240 run bench_spans20 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --spans 20
L=3 T=240 run bench_spans8 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --spans 8
