#!/bin/bash
# round 6: the predict record with its kernel trace, the new one-rank RCCL test, the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "rccl" 2>&1 | grep -i "passed\|failed\|error" | tail -4 | cut -c1-400
cat > /tmp/predict.py <<'P'
import json, sys
sys.path.insert(0, ".")
import torch, bench
print(json.dumps(bench.c3_predict_line(torch.device("cuda:0"))))
P
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python /tmp/predict.py > $O/predict.log 2>&1
db=$(find $O/prof -name '*.db' | head -1)
python tools/rocprof_summary.py stats "$db" > $O/kernel_stats_c3_predict.txt 2>&1
rm -rf $O/prof; grep c3-svgp $O/predict.log | cut -c1-900; head -16 $O/kernel_stats_c3_predict.txt | cut -c1-70,100-200
