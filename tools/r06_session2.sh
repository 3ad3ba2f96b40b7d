#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s2
( nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"; cat /proc/loadavg ) > gpurun_out/s2/cpus.txt 2>&1
L="devlibs/base.so devlibs/noties.so devlibs/notieswait3.so devlibs/abl3.so devlibs/wait3.so"
for inp in pcg noise; do
  echo "== input $inp" >> gpurun_out/s2/ab.txt
  AB_INPUT=$inp AB_ROUNDS=7 AB_STEPS=20 timeout 600 python tools/ab_bench.py $L >> gpurun_out/s2/ab.txt 2>&1
done
cat gpurun_out/s2/cpus.txt; cat gpurun_out/s2/ab.txt
