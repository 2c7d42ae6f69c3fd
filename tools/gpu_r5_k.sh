#!/bin/bash
# round 5: the fused RBF reverse kernel -- gradient tests, the round-5 sweep fixtures, kernel statistics and counters of K(X) forward + backward
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05k; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_grad.py -q -x > $O/pytest_gpu_grad.log 2>&1; tail -3 $O/pytest_gpu_grad.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "round5 or float32_requests or knowingly" > $O/pytest_fixtures.log 2>&1; tail -3 $O/pytest_fixtures.log
export TMPDIR=/tmp; rm -rf /tmp/pmc_run
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pmc_run -o p -- python tools/bench_grad_gram.py 1024 rbf 5 > $O/prof_grad.log 2>&1
tail -2 $O/prof_grad.log
db=$(find /tmp/pmc_run -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" | cut -c1-250 > $O/kernel_stats_grad_rbf.txt 2>&1
head -12 $O/kernel_stats_grad_rbf.txt | cut -c1-60,118-200
for set in "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
  rm -rf /tmp/pmc_run
  timeout 600 rocprofv3 --pmc $set -d /tmp/pmc_run -o p -- python tools/bench_grad_gram.py 1024 rbf 3 > $O/pmc.log 2>&1
  db=$(find /tmp/pmc_run -name '*.db' | head -1)
  echo "== --pmc $set" >> $O/pmc_grad_rbf.txt
  python tools/rocprof_summary.py pmc "$db" seq_grad_fused 2>&1 | head -4 | cut -c1-260 >> $O/pmc_grad_rbf.txt
done
cat $O/pmc_grad_rbf.txt | cut -c1-200
