#!/bin/bash
# round 5: the stash route -- full GPU suite, default bench, kernel statistics and counters of K(X) forward + backward
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05x; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json
timeout 600 python tools/gpu_stash_check.py > $O/stash_check.txt 2>&1; tail -8 $O/stash_check.txt
export TMPDIR=/tmp; rm -rf /tmp/pmc_run
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pmc_run -o p -- python tools/bench_grad_gram.py 1024 rbf 5 > $O/prof_grad.log 2>&1
db=$(find /tmp/pmc_run -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" | cut -c1-250 > $O/kernel_stats_grad_rbf.txt 2>&1
head -6 $O/kernel_stats_grad_rbf.txt | cut -c1-70,118-200
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES"; do
  rm -rf /tmp/pmc_run
  timeout 600 rocprofv3 --pmc $set -d /tmp/pmc_run -o p -- python tools/bench_grad_gram.py 1024 rbf 3 > $O/pmc.log 2>&1
  db=$(find /tmp/pmc_run -name '*.db' | head -1)
  echo "== --pmc $set" >> $O/pmc_grad_rbf.txt
  python tools/rocprof_summary.py pmc "$db" seq_grad_fused 2>&1 | head -3 | cut -c1-260 >> $O/pmc_grad_rbf.txt
  python tools/rocprof_summary.py pmc "$db" seq_gram_kernel 2>&1 | head -3 | tail -2 | cut -c1-260 >> $O/pmc_grad_rbf.txt
done
cat $O/pmc_grad_rbf.txt | cut -c1-200
