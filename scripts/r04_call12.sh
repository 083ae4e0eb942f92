#!/bin/bash
# GPU call 12: self-attention step (long-context loads batched) bit-identity + effect on the sequential / span modes; RCCL two-ranks-one-GPU probe
mkdir -p gpurun_out
B="--no-cpu-baseline --no-f32 --no-roofline"
echo "== self-attn step check"; timeout 200 python tests/hw_checks/self_attn_step_check.py 2>&1 | tail -10
echo "== decode tests"; timeout 600 python -m pytest tests/test_gpu_model.py -q --timeout=500 --tb=short -k "graph_replay or fused_cross_query or batched_windows" 2>&1 | tail -4
echo "== rccl 2 ranks on 1 GPU"; timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 scripts/rccl_two_ranks_one_gpu.py 2>&1 | grep -v "^W0\|^\*\*\*\|OMP_NUM" | tail -6 | cut -c1-500 | tee gpurun_out/r04_rccl_two_ranks_one_gpu.txt
echo "== sequential"; ( timeout 400 python bench.py --sequential --steps 1 --warmup 1 $B 2>&1 | tail -1 ) | tee gpurun_out/r04_c12_bench_sequential.json | cut -c1-260
echo "== spans 20"; ( timeout 300 python bench.py --spans 20 --steps 2 --warmup 1 $B 2>&1 | tail -1 ) | tee gpurun_out/r04_c12_bench_spans20.json | cut -c1-260
echo "== headline"; ( timeout 300 python bench.py --steps 5 --warmup 2 $B 2>&1 | tail -1 ) | tee gpurun_out/r04_c12_bench_default.json | cut -c1-260
