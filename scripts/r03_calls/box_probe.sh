#!/bin/bash
# what kind of box is this?  rocm-smi state + the headline command (short) + the same with the host's launch thread pinned
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_boxprobe_$(date +%s)
mkdir -p $O
export PYTHONUNBUFFERED=1
( rocm-smi --showcomputepartition --showmemorypartition --showclocks --showpower --showperflevel --showtemp --showmemuse 2>&1 | head -80 ) > $O/smi_before.txt
( rocminfo 2>/dev/null | grep -E "Compute Unit|Max Clock|Cacheline|Marketing|Wavefront|L2|L3|Name:" | head -40 ) > $O/rocminfo.txt
( lscpu | head -25; cat /sys/fs/cgroup/cpu.max; uname -r; cat /proc/meminfo | grep -i -E "hugepages|MemTotal" ) > $O/host.txt 2>&1
timeout 300 python bench.py --steps 4 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline > $O/bench.json 2>> $O/bench.err
( rocm-smi --showclocks --showpower --showperflevel 2>&1 | head -40 ) > $O/smi_after.txt
python -c "import json; j=json.load(open('$O/bench.json')); print('ms_per_step', j['ms_per_step'])"
grep -E "sclk|mclk|fclk|socclk|Power|Perf|partition|Partition" $O/smi_before.txt | head -20
