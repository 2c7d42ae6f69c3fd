#!/bin/bash
# round 5: full GPU suite + the default bench line with the fused reverse kernel in the library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05l; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
