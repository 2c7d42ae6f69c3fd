#!/bin/bash
# Round 5's GPU-box visit: parity tests, the default bench line, every bench configuration, rocprofv3 kernel stats and the PMC passes (each counter
# set in its own pass, no tracing: MI355X_MICROARCH.md) for each, the tile kernel's variants and its two-level-table A/B with counters, gradients.
# Output under gpurun_out/r05.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -3 $O/pytest_gpu.log
fi
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
: > $O/bench_lines.jsonl
for cfg in "c2" "c2 --lattice" "c2 --base rbf" "c3" "c3 --increments" "c3 --base linear" "c5" "c5 --base linear"; do
  timeout 900 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench_lines.jsonl
done
timeout 900 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench_lines.jsonl
for cfg in "c2" "c2 --lattice" "c2 --base rbf" "c3" "c3 --increments" "c5"; do
  tag=$(echo $cfg | tr -d ' -')
  : > $O/pmc_$tag.txt
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64"; do
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
# the two-level (bank-conflict-free) exp table against the 1,024-entry one: times alternating, then the LDS counters of the variant
if [ -f gpsig_amd/lib/libgpsig_hip_e32.so ]; then
  AB_CFGS="c3 --increments" AB_STEPS=20 bash tools/gpu_ab_libs.sh libgpsig_hip_e32.so > $O/ab_c3incr_e32.txt 2>&1
  rm -rf /tmp/pmc_run
  GPSIG_LIB=$PWD/gpsig_amd/lib/libgpsig_hip_e32.so timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES -d /tmp/pmc_run -o p -- python bench.py --config c3 --increments --steps 2 --warmup 1 --timed-loop-only > $O/pmc_run_e32.log 2>&1
  db=$(find /tmp/pmc_run -name '*.db' | head -1)
  (echo "## libgpsig_hip_e32.so (TVS_EXPTAB=32: two conflict-free tables of 32 entries): rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES (bench.py --config c3 --increments)"; python tools/rocprof_summary.py pmc "$db" gpsig 2>&1 | awk '$3 > 200 || NR == 1' | cut -c1-260) >> $O/ab_c3incr_e32.txt
  rm -rf /tmp/pmc_run
fi
timeout 600 python tools/bench_c3.py > $O/bench_c3_variants.txt 2>&1
timeout 600 python tools/bench_grad.py > $O/bench_grad.txt 2>&1
rm -rf /tmp/pmc_run
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pmc_run -o p -- python tools/bench_grad_gram.py 1024 rbf 5 > $O/prof_grad_rbf.log 2>&1
db=$(find /tmp/pmc_run -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" | cut -c1-250 > $O/kernel_stats_grad_rbf.txt 2>&1
rm -rf /tmp/pmc_run
for cfg in "c3 --verify" "c2 --verify"; do timeout 600 python tools/bench_lr.py --config $cfg 2>/dev/null >> $O/bench_lowrank.jsonl; done
timeout 300 python tools/bench_host_e2e.py > $O/bench_host_e2e.txt 2>&1
timeout 600 python tools/bench_rank_share.py > $O/bench_rank_share.txt 2>&1
GPSIG_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_2ranks_gloo_one_gpu.json
timeout 300 tools/microbench4 20000 > $O/microbench4.txt 2>&1
ls $O
