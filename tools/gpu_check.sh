#!/bin/bash
# Quick GPU visit: the whole GPU test-suite, then bench lines for the configs named on the command line (default: all).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
for cfg in "$@"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout 600 python bench.py --config $cfg --no-cpu-baseline > $O/q_bench_$tag.json 2> $O/q_bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/q_bench_$tag.json").read().strip().splitlines()[-1])
    print("$cfg", "ms/step", round(d["ms_per_step"],3), "kernel ms", round(d["roofline"]["kernel_ms_per_launch"],3), "frac", round(d["roofline"]["frac"],3), "alu", round(d["roofline"]["alu"]["alu_frac"],3), "rel_err", d["rel_err"])
except Exception as e:
    print("$cfg FAILED", e); print(open("$O/q_bench_$tag.err").read()[-2000:])
PY
done
