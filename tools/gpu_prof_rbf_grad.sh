#!/bin/bash
# kernel stats of K(X) forward + backward, RBF, N=1024 at the headline shape (pair kernels' reverse pass)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_rbf
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_rbf -o p -- python tools/bench_grad_gram.py 1024 rbf 5 > /dev/null 2>&1
db=$(find /tmp/prof_rbf -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" | cut -c1-250 > gpurun_out/kernel_stats_grad_gram_rbf.txt 2>&1
head -16 gpurun_out/kernel_stats_grad_gram_rbf.txt | cut -c1-80,112-250
