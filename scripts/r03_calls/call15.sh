#!/bin/bash
# round 3, call 15: does the work in front of a one-window encoder pass change its duration? (align(): 1.35x the isolated loop)
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_c15
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
echo "host kernel $(uname -r)" | tee $O/box.txt
timeout 400 python scripts/exp/encode_b1_timing.py 2>&1 | grep -v amdgpu.ids > $O/encode_b1_timing.txt; cat $O/encode_b1_timing.txt
timeout 400 python scripts/exp/align_encode_probe.py 2>&1 | grep -v amdgpu.ids > $O/align_encode_probe.txt; cat $O/align_encode_probe.txt
