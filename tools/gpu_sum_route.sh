#!/bin/bash
# the level sum's gradient as one op (gpsig_kernel_K_grad) against the level primitives + torch ops
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for args in "1024 linear 10" "4096 linear 5" "4096 linear 5 100 6 4" "1024 linear 10 100 6 4" "2048 linear 10 64 16 3" "1024 linear 10 50 3 4"; do
  GPSIG_SUM_ROUTE=0 python tools/bench_grad_gram.py $args 2>/dev/null | tail -1
  python tools/bench_grad_gram.py $args 2>/dev/null | tail -1
done
} | tee gpurun_out/bench_sum_route.txt
export TMPDIR=/tmp
rm -rf /tmp/prof_sum
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sum -o p -- python tools/bench_grad_gram.py 4096 linear 5 > /dev/null 2>&1
db=$(find /tmp/prof_sum -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" | cut -c1-250 > gpurun_out/kernel_stats_grad_gram_sum.txt 2>&1
head -24 gpurun_out/kernel_stats_grad_gram_sum.txt | cut -c1-60,112-250
