#!/bin/bash
# round 5: HBM traffic of the fused reverse kernel (FETCH_SIZE / WRITE_SIZE in passes of their own) + a randomised gradient sweep
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05m; mkdir -p $O; export TMPDIR=/tmp
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_run
  timeout 600 rocprofv3 --pmc $set -d /tmp/pmc_run -o p -- python tools/bench_grad_gram.py 1024 rbf 3 > $O/pmc.log 2>&1
  db=$(find /tmp/pmc_run -name '*.db' | head -1)
  echo "== --pmc $set" >> $O/pmc_traffic_grad_rbf.txt
  python tools/rocprof_summary.py pmc "$db" seq_grad_fused 2>&1 | head -4 | cut -c1-260 >> $O/pmc_traffic_grad_rbf.txt
done
cat $O/pmc_traffic_grad_rbf.txt
timeout 1500 python tools/fuzz_grad.py 400 91 > $O/fuzz_grad_91.txt 2>&1; tail -4 $O/fuzz_grad_91.txt
FUZZ_ORDER=1 timeout 900 python tools/fuzz_grad.py 150 92 > $O/fuzz_grad_92_orders.txt 2>&1; tail -3 $O/fuzz_grad_92_orders.txt
