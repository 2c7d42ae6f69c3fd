#!/bin/bash
# round 5: bench test + the fused kernel at the full headline size
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bench.py -q -x > $O/pytest_bench.log 2>&1; tail -3 $O/pytest_bench.log
timeout 600 python tools/bench_grad_gram.py 4096 rbf 2 2>&1 | tail -1 > $O/grad_n4096.txt; cat $O/grad_n4096.txt
