#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_c3l
BENCH_GRAD_BASES=linear BENCH_GRAD_INCR=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3l -o p -- python tools/bench_grad.py b > /dev/null 2>&1
db=$(find /tmp/prof_c3l -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" | cut -c1-250 > gpurun_out/kernel_stats_c3lin_features.txt 2>&1
head -30 gpurun_out/kernel_stats_c3lin_features.txt | cut -c1-90,112-250
