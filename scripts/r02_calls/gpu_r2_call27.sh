#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) ; nproc $(nproc) ; cfs: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat 2>/dev/null
for cfg in "OMP_NUM_THREADS=1" "OMP_NUM_THREADS=8 OMP_WAIT_POLICY=PASSIVE GOMP_SPINCOUNT=0" "A=1"; do
  echo "=== $cfg"; ( env $cfg timeout 300 python scripts/micro/align_trace.py 2>&1 | tail -5 | cut -c1-420 )
  grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null
  echo "== align $cfg"; ( env $cfg timeout 600 python bench.py --mode align --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
done 2>&1 | tee gpurun_out/omp_ab.txt
echo "== bench OMP_NUM_THREADS=1"; ( OMP_NUM_THREADS=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 --no-roofline 2>&1 | tail -1 | cut -c1-330 )
