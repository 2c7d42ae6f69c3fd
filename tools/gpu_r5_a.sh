#!/bin/bash
# Round 5, first GPU visit: (1) instruction issue costs (tools/microbench4), (2) the Kzx tile kernel's rewrite (persistent item queue, rows by
# scalar loads, hand-scheduled exp pairs, three waves per SIMD) -- parity tests of everything that goes through it, then configs[2] A/B against the
# round-4 library and the two-level exp table variant, (3) its counters.  Output under gpurun_out/r05a.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${R5_OUT:-r05a}; mkdir -p $O
export TMPDIR=/tmp
timeout 300 tools/microbench4 20000 > $O/microbench4.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tile or tens or config3 or golden or covs or notebook or lane" > $O/pytest_tile.log 2>&1; echo "rc=$?" >> $O/pytest_tile.log
tail -3 $O/pytest_tile.log
for rnd in 1 2 3; do
  for cfg in "c3" "c3 --increments"; do
    for lib in default libgpsig_hip_r4.so ${AB_LIBS:-}; do
      if [ $lib = default ]; then unset GPSIG_LIB; else export GPSIG_LIB=$PWD/gpsig_amd/lib/$lib; fi
      [ $lib = default ] || [ -f gpsig_amd/lib/$lib ] || continue
      timeout 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $rnd  $cfg  lib=$lib  kernel ms %.3f  ms/step %.3f  rel_err %.2e  clock %.2f' % (d['roofline']['kernel_ms_per_launch'], d['ms_per_step'], d['rel_err'], d['clock_ghz']))" >> $O/ab_c3.txt 2>&1
    done
  done
done
unset GPSIG_LIB
cat $O/ab_c3.txt
for cfg in "c3" "c3 --increments"; do
  tag=$(echo $cfg | tr -d ' -')
  : > $O/pmc_$tag.txt
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pmc_run
    timeout 600 rocprofv3 --pmc $set -d /tmp/pmc_run -o p -- python bench.py --config $cfg --steps 2 --warmup 1 --timed-loop-only > $O/pmc_run_$tag.log 2>&1
    db=$(find /tmp/pmc_run -name '*.db' | head -1)
    echo "## rocprofv3 --pmc $set   (bench.py --config $cfg --steps 2 --warmup 1 --timed-loop-only)" >> $O/pmc_$tag.txt
    python tools/rocprof_summary.py pmc "$db" "${PMC_FILTER:-gpsig}" 2>&1 | awk '$3 > 200 || NR == 1' | cut -c1-260 >> $O/pmc_$tag.txt
    echo >> $O/pmc_$tag.txt
  done
  rm -rf /tmp/pmc_run
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pmc_run -o p -- python bench.py --config $cfg --steps 5 --warmup 2 --timed-loop-only > $O/prof_$tag.log 2>&1
  db=$(find /tmp/pmc_run -name '*.db' | head -1)
  python tools/rocprof_summary.py stats "$db" | cut -c1-250 > $O/kernel_stats_$tag.txt 2>&1
  rm -rf /tmp/pmc_run
done
timeout 600 python tools/bench_grad.py b > $O/bench_grad_b.txt 2>&1
tail -5 $O/bench_grad_b.txt
