#!/bin/bash
# round 6, final measurement pass with the round's library: bench default (all records), kernel trace + PMC passes of the headline command,
# the order benches, the reference-shapes visit (tools/gpu_r6_h.sh), and the GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json
bash tools/gpu_pmc.sh c2 > $O/pmc.log 2>&1
cp gpurun_out/r02/pmc_c2.txt gpurun_out/r02/kernel_stats_c2.txt $O/ 2>/dev/null
python tools/bench_order_rbf.py > $O/bench_order_rbf.txt 2>&1
python tools/bench_order_kzx.py > $O/bench_order_kzx.txt 2>&1
bash tools/gpu_r6_h.sh > $O/shapes.log 2>&1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6 > $O/pytest_gpu.txt
tail -2 $O/pytest_gpu.txt
