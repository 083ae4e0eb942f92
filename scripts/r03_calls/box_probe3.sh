#!/bin/bash
# box probe 3: what a node is configured with (amdgpu module parameters, kernel command line, KFD node properties, firmware)
# next to a 10-second classification of the box (decode-GEMM micro-benchmark: normal boxes 5.8-9.7 us per launch, slow ones +6-8)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_boxprobe3_$(date +%s)
mkdir -p $O
export PYTHONUNBUFFERED=1
echo "host kernel $(uname -r)" | tee $O/kernel.txt
( for f in /sys/module/amdgpu/parameters/*; do echo "$(basename $f)=$(cat $f 2>/dev/null)"; done ) > $O/amdgpu_params.txt 2>&1
cat /proc/cmdline > $O/cmdline.txt 2>&1
( for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "== $n"; cat $n/properties 2>/dev/null | grep -E "simd_count|cu_count|max_waves|lds_size|gfx_target|sdma|num_xcc|debug_prop|capability|fw_version|max_engine_clk|unique_id" ; done ) > $O/kfd_nodes.txt 2>&1
( cat /sys/module/amdgpu/version 2>/dev/null; rocm-smi --showfwinfo 2>/dev/null | grep -E "MEC|SMC|RLC|SDMA|VBIOS|PSP|TA|IMU" | head -20; rocm-smi --showmemuse --showperflevel 2>/dev/null | grep -E "GPU\[0\]" | head -6 ) > $O/fw.txt 2>&1
( env | grep -E "^HSA_|^HIP_|^ROC|^GPU_|^AMD" ) > $O/env.txt 2>&1
timeout 120 python scripts/kernel_bench.py --only dec --iters 100 2>&1 | grep -v amdgpu.ids > $O/kb_dec.txt
tail -8 $O/kb_dec.txt
grep -E "^(mtype_local|sched_policy|hws_|noretry|cwsr|mes|queue_preemption|halt_if|vm_|svm_|sdma_phase|num_kcq|tmz|timeout)" $O/amdgpu_params.txt | tr '\n' ' '
