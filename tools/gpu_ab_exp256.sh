#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r02
bash tools/ab_lib.sh libgpsig_hip_exp256.so "c3" "c3 --increments" > gpurun_out/r02/ab_exp256.txt 2>&1
python bench.py --config c2 --steps 5 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['cpu_baseline']
print('cpu_baseline C port', c['value'], c['cores'], c['sample']); print('numpy', c['numpy'])" >> gpurun_out/r02/ab_exp256.txt 2>&1
python bench.py --config c3 --steps 5 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['cpu_baseline']
print('cpu_baseline C port', c['value'], c['cores'], c['sample']); print('numpy', c['numpy'])" >> gpurun_out/r02/ab_exp256.txt 2>&1
