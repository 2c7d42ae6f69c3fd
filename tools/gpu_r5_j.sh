#!/bin/bash
# round 5: full GPU suite on the deduplicated build + the Lambda-scratch experiment of the RBF reverse pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for mb in 4096 1024 256 128 64; do
  GPSIG_OPTIONS="grad_scratch_mb=$mb" timeout 300 python tools/bench_grad_gram.py 1024 rbf 5 2>&1 | tail -1 >> $O/grad_scratch.txt
done
cat $O/grad_scratch.txt
export TMPDIR=/tmp; rm -rf /tmp/pmc_run
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pmc_run -o p -- python tools/bench_grad_gram.py 1024 rbf 5 > $O/prof_grad.log 2>&1
db=$(find /tmp/pmc_run -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" | cut -c1-250 > $O/kernel_stats_grad_rbf.txt 2>&1
head -12 $O/kernel_stats_grad_rbf.txt | cut -c1-60,118-200
