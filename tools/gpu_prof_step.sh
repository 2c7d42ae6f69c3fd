cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/grad; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python tools/bench_grad.py a > $O/prof.log 2>&1
db=$(find $O/prof -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" > $O/kernel_stats_step.txt 2>&1
rm -rf $O/prof; tail -3 $O/prof.log; wc -l $O/kernel_stats_step.txt
