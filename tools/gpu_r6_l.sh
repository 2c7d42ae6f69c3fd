#!/bin/bash
# one kernel trace of a probe script: bash tools/gpu_r6_l.sh <name> <script> [args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06l; mkdir -p $O
name=$1; shift
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o p -- python "$@" > $O/run_$name.log 2>&1
db=$(find $O/prof_$name -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" > $O/kernels_$name.txt 2>&1
rm -rf $O/prof_$name
head -24 $O/kernels_$name.txt | cut -c1-250
