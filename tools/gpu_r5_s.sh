#!/bin/bash
# round 5, closing run: full GPU suite, default bench, a forward parity sweep with the final library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05s; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
timeout 1300 python tools/fuzz_parity.py 800 73 > $O/fuzz_parity_73.txt 2>&1; tail -4 $O/fuzz_parity_73.txt | cut -c1-300
