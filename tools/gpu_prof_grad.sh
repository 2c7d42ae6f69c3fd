#!/bin/bash
# rocprofv3 kernel stats of tools/bench_grad.py b (forward + backward of the three SVGP covariances at BASELINE configs[2]).
# usage: tools/gpu_prof_grad.sh [tag]   ->  gpurun_out/r03/kernel_stats_grad_c3[_tag].txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
tag=${1:+_$1}
rm -rf /tmp/pg; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pg -o p -- python tools/bench_grad.py b > $O/prof_grad$tag.log 2>&1
db=$(find /tmp/pg -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" | cut -c1-250 | head -40 > $O/kernel_stats_grad_c3$tag.txt
