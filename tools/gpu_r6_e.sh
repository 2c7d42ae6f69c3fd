#!/bin/bash
# round 6: kernel trace of one SVGP step at two of the reference's shapes (wide route), + the sweep of the wide data sets
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06e; mkdir -p $O
for ds in ${DATASETS:-NetFlow CMUsubject16}; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$ds -o p -- python tools/reference_shapes.py $ds --routes auto --reps 3 > $O/prof_$ds.log 2>&1
  db=$(find $O/prof_$ds -name '*.db' | head -1)
  python tools/rocprof_summary.py stats "$db" > $O/kernel_stats_$ds.txt 2>&1
  rm -rf $O/prof_$ds; grep dataset $O/prof_$ds.log | cut -c1-700; head -25 $O/kernel_stats_$ds.txt | cut -c1-60,100-200
done
