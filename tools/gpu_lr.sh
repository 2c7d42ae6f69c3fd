#!/bin/bash
# Low-rank mode on the GPU box: parity tests, bench_lr at both shapes with and without the fused feature kernel, kernel stats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/lr; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "low_rank or tensor_gram or diagonal_pass" > $O/pytest_lr.log 2>&1; echo "pytest rc=$?" >> $O/pytest_lr.log
tail -5 $O/pytest_lr.log
for cfg in "c3 --fused 1 --verify" "c3 --fused 2" "c3 --fused 0" "c3 --fused 1 --sparsity log" "c3 --fused 1 --sparsity lin" "c3 --fused 1 --base linear --verify" "c2 --fused 1 --verify" "c2 --fused 0" "c3 --fused 1 --components 100" "c3 --fused 0 --components 100"; do
  timeout 600 python tools/bench_lr.py --config $cfg 2>$O/err.log | tee -a $O/bench_lr.jsonl | cut -c1-900; tail -3 $O/err.log
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python tools/bench_lr.py --config c3 --fused 1 --steps 5 > $O/prof.log 2>&1
db=$(find $O/prof -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" > $O/kernel_stats_lr_c3.txt 2>&1 || true
rm -rf $O/prof
head -30 $O/kernel_stats_lr_c3.txt | cut -c1-220
# PMC passes of the fused feature kernel (each counter set in its own pass, no tracing)
: > $O/pmc_lr_c3.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/pmc_run
  timeout 600 rocprofv3 --pmc $set -d /tmp/pmc_run -o p -- python tools/bench_lr.py --config c3 --steps 2 > $O/pmc_run.log 2>&1
  db=$(find /tmp/pmc_run -name '*.db' | head -1)
  echo "## rocprofv3 --pmc $set   (tools/bench_lr.py --config c3 --steps 2)" >> $O/pmc_lr_c3.txt
  python tools/rocprof_summary.py pmc "$db" "lr_seq_features_fused" 2>&1 | cut -c1-260 >> $O/pmc_lr_c3.txt
  echo >> $O/pmc_lr_c3.txt
done
rm -rf /tmp/pmc_run
cat $O/pmc_lr_c3.txt | cut -c1-230
