#!/bin/bash
# box probe 2: headline command (short) with sclk / power sampled during the decode phase; on a slow box (> 520 ms) the same again
# with the performance level forced high, then restored
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03_boxprobe2_$(date +%s)
mkdir -p $O
export PYTHONUNBUFFERED=1
( uname -r; cat /sys/module/amdgpu/version 2>/dev/null; rocm-smi --showdriverversion --showfwinfo 2>/dev/null | grep -E "Driver|firmware|MEC|SMC|RLC|SDMA|VBIOS|PSP|TA" | head -30 ) > $O/versions.txt 2>&1
echo "host kernel $(uname -r)"
sample() { for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|fclk|Power" | tr '\n' ' ' ; echo; sleep 0.25; done; }
sample > $O/clocks_run1.txt &
SP=$!
timeout 300 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline > $O/bench1.json 2>> $O/bench.err
kill $SP 2>/dev/null
MS=$(python -c "import json; print(json.load(open('$O/bench1.json'))['ms_per_step'])")
echo "run1 ms_per_step $MS"
sort $O/clocks_run1.txt | uniq -c | sort -rn | head -4
if python -c "import sys; sys.exit(0 if float('$MS') > 520 else 1)"; then
  echo "SLOW BOX: forcing perf level high"
  rocm-smi --setperflevel high > $O/setperf.txt 2>&1; tail -2 $O/setperf.txt
  sample > $O/clocks_run2.txt &
  SP=$!
  timeout 300 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline > $O/bench2.json 2>> $O/bench.err
  kill $SP 2>/dev/null
  python -c "import json; print('run2 (perf level high) ms_per_step', json.load(open('$O/bench2.json'))['ms_per_step'])"
  sort $O/clocks_run2.txt | uniq -c | sort -rn | head -4
  rocm-smi --setperflevel auto > /dev/null 2>&1
  timeout 300 python bench.py --steps 6 --warmup 2 --no-f32 --no-cpu-baseline --no-roofline > $O/bench3.json 2>> $O/bench.err
  python -c "import json; print('run3 (auto again) ms_per_step', json.load(open('$O/bench3.json'))['ms_per_step'])"
fi
