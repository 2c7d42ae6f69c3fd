#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O; export TMPDIR=/tmp
rm -rf /tmp/pg; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pg -o p -- python tools/bench_grad.py b > $O/prof_grad.log 2>&1
db=$(find /tmp/pg -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" | cut -c1-250 | head -30 > $O/kernel_stats_grad_c3.txt
