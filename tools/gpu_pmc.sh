#!/bin/bash
# rocprofv3 PMC passes (each counter set in its own pass, no tracing: MI355X_MICROARCH.md) + kernel-trace stats for the bench
# configurations named on the command line.  Output: gpurun_out/r02/pmc_<tag>.txt, kernel_stats_<tag>.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
for cfg in "$@"; do
  tag=$(echo $cfg | tr -d ' -')
  : > $O/pmc_$tag.txt
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
    rm -rf /tmp/pmc_run
    timeout 600 rocprofv3 --pmc $set -d /tmp/pmc_run -o p -- python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_run_$tag.log 2>&1
    db=$(find /tmp/pmc_run -name '*.db' | head -1)
    echo "## rocprofv3 --pmc $set   (bench.py --config $cfg --steps 2 --warmup 1)" >> $O/pmc_$tag.txt
    python tools/rocprof_summary.py pmc "$db" "${PMC_FILTER:-gpsig}" 2>&1 | awk '$3 > 200 || NR == 1' | cut -c1-260 >> $O/pmc_$tag.txt
    echo >> $O/pmc_$tag.txt
  done
  rm -rf /tmp/pmc_run
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pmc_run -o p -- python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_$tag.log 2>&1
  db=$(find /tmp/pmc_run -name '*.db' | head -1)
  python tools/rocprof_summary.py stats "$db" | cut -c1-250 > $O/kernel_stats_$tag.txt 2>&1
  rm -rf /tmp/pmc_run
done
