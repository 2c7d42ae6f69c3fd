#!/bin/bash
# Low-rank mode on the GPU box: parity tests, bench_lr at both shapes with and without the fused feature kernel, kernel stats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/lr; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "low_rank or tensor_gram or diagonal_pass" > $O/pytest_lr.log 2>&1; echo "pytest rc=$?" >> $O/pytest_lr.log
tail -5 $O/pytest_lr.log
for cfg in "c3 --fused 1 --verify" "c3 --fused 0" "c3 --fused 1 --sparsity log" "c3 --fused 1 --sparsity lin" "c3 --fused 1 --base linear --verify" "c2 --fused 1 --verify" "c2 --fused 0" "c3 --fused 1 --components 100" "c3 --fused 0 --components 100"; do
  timeout 600 python tools/bench_lr.py --config $cfg 2>$O/err.log | tee -a $O/bench_lr.jsonl | cut -c1-900; tail -3 $O/err.log
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python tools/bench_lr.py --config c3 --fused 1 --steps 5 > $O/prof.log 2>&1
db=$(find $O/prof -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" > $O/kernel_stats_lr_c3.txt 2>&1 || true
rm -rf $O/prof
head -30 $O/kernel_stats_lr_c3.txt | cut -c1-220
