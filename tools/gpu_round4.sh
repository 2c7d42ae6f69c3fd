#!/bin/bash
# Round 4's GPU-box visit: parity tests, the default bench line, every bench configuration, rocprofv3 kernel stats and the PMC
# passes (each counter set in its own pass, no tracing: MI355X_MICROARCH.md) for each.  Output under gpurun_out/r04.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -3 $O/pytest_gpu.log
fi
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
for cfg in "c2 --lattice" "c2 --base rbf" "c3" "c3 --increments" "c3 --base linear" "c5" "c5 --base linear"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout 600 python bench.py --config $cfg --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
done
timeout 900 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_1gpu.json 2> $O/bench_c4_1gpu.err
for cfg in "c2" "c2 --lattice" "c2 --base rbf" "c3" "c3 --increments" "c3 --base linear" "c5"; do
  tag=$(echo $cfg | tr -d ' -')
  : > $O/pmc_$tag.txt
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64"; do
    case "$set" in *MFMA*) [ "$tag" = "c2" ] || continue;; esac
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
for cfg in "c3 --verify" "c2 --verify"; do timeout 600 python tools/bench_lr.py --config $cfg 2>/dev/null >> $O/bench_lowrank.jsonl; done
timeout 600 python tools/bench_grad.py > $O/bench_grad.txt 2>&1
timeout 300 python tools/bench_host_e2e.py > $O/bench_host_e2e.txt 2>&1
timeout 600 python tools/bench_rank_share.py > $O/bench_rank_share.txt 2>&1
timeout 300 bash tools/gpu_sum_route.sh > $O/sum_route.log 2>&1
BENCH_GRAD_BASES=linear GPSIG_FEATURE_ROUTE=0 timeout 300 python tools/bench_grad.py b 2>&1 | grep "(b)" > $O/bench_grad_c3_linear_recursion.txt
