#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
python tools/prof_draw.py sqrt lin 2>&1 | grep -v amdgpu.ids
rm -rf /tmp/pd; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pd -o p -- python tools/prof_draw.py > $O/prof_draw.log 2>&1
db=$(find /tmp/pd -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" | cut -c1-200 | head -20 | tee $O/kernel_stats_lr_draw.txt
