#!/bin/bash
# rocprofv3 PMC passes (no tracing) of tools/bench_grad.py b for one (base, increments): BENCH_GRAD_BASES=rbf BENCH_GRAD_INCR=0 tools/gpu_pmc_grad.sh tag
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
tag=${1:-grad}
: > $O/pmc_$tag.txt
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_run
  timeout 600 rocprofv3 --pmc $set -d /tmp/pmc_run -o p -- python tools/bench_grad.py b > $O/pmc_run_$tag.log 2>&1
  db=$(find /tmp/pmc_run -name '*.db' | head -1)
  echo "## rocprofv3 --pmc $set   (tools/bench_grad.py b, BENCH_GRAD_BASES=${BENCH_GRAD_BASES:-} BENCH_GRAD_INCR=${BENCH_GRAD_INCR:-})" >> $O/pmc_$tag.txt
  python tools/rocprof_summary.py pmc "$db" "${PMC_FILTER:-tvs_grad_tile_kernel}" 2>&1 | cut -c1-300 >> $O/pmc_$tag.txt
  echo >> $O/pmc_$tag.txt
done
rm -rf /tmp/pmc_run
