#!/bin/bash
# round 6: kernel trace + counter passes (each counter set in its own pass, no tracing: MI355X_MICROARCH.md) of the order-2 reverse pass (tools/probe_ho_grad.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06m; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ho_tr -o p -- python tools/probe_ho_grad.py 2 > $O/trace.log 2>&1
python tools/rocprof_summary.py stats "$(find /tmp/ho_tr -name '*.db' | head -1)" | head -14 | cut -c1-250 > $O/kernel_stats_grad_n512_rbf_order2.txt
: > $O/pmc_grad_n512_rbf_order2.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/ho_pmc
  timeout 600 rocprofv3 --pmc $set -d /tmp/ho_pmc -o p -- python tools/probe_ho_grad.py 2 > $O/pmc.log 2>&1
  echo "## rocprofv3 --pmc $set   (python tools/probe_ho_grad.py 2)" >> $O/pmc_grad_n512_rbf_order2.txt
  python tools/rocprof_summary.py pmc "$(find /tmp/ho_pmc -name '*.db' | head -1)" gpsig 2>&1 | awk '$3 > 200 || NR == 1' | cut -c1-260 >> $O/pmc_grad_n512_rbf_order2.txt
  echo >> $O/pmc_grad_n512_rbf_order2.txt
done
cat $O/kernel_stats_grad_n512_rbf_order2.txt | cut -c1-60,118-210
cat $O/pmc_grad_n512_rbf_order2.txt | cut -c1-230
