#!/bin/bash
# One GPU-box visit: parity tests, every bench configuration, rocprofv3 kernel stats for each.  Output under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
for cfg in "c2" "c2 --base rbf" "c3" "c3 --increments" "c3 --base linear" "c5" "c5 --base linear"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout 600 python bench.py --config $cfg > $O/bench_$tag.json 2> $O/bench_$tag.err; tail -1 $O/bench_$tag.json
done
timeout 900 python bench.py --config c4 --steps 2 --warmup 1 > $O/bench_c4_1gpu.json 2> $O/bench_c4_1gpu.err; tail -1 $O/bench_c4_1gpu.json
for cfg in "c2" "c3" "c3 --increments" "c5"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_$tag.log 2>&1
  db=$(find $O/prof_$tag -name '*.db' | head -1)
  python tools/rocprof_summary.py stats "$db" > $O/kernel_stats_$tag.txt 2>&1 || true
  rm -rf $O/prof_$tag
done
# low-rank mode and the spectral kernel (tools of their own: their natural yard-sticks are not the pair-stream roofline)
for cfg in "c3 --verify" "c2 --verify"; do timeout 600 python tools/bench_lr.py --config $cfg 2>/dev/null >> $O/bench_lowrank.jsonl; done
timeout 600 python tools/bench_spectral.py > $O/bench_spectral.txt 2>&1
