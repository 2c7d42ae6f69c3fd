#!/bin/bash
# round 6, closing sweeps with fresh seeds on the final library (larger counts than tools/gpu_r6_k.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06n; mkdir -p $O
timeout 2400 python tools/fuzz_parity.py 1500 91 > $O/fuzz_parity_91.txt 2>&1; grep "cases" $O/fuzz_parity_91.txt
FUZZ_WIDE=1 FUZZ_ORDER=1 timeout 2400 python tools/fuzz_parity.py 600 92 > $O/fuzz_parity_wide_orders_92.txt 2>&1; grep "cases" $O/fuzz_parity_wide_orders_92.txt
FUZZ_ORDER=1 timeout 2400 python tools/fuzz_grad.py 500 93 > $O/fuzz_grad_orders_93.txt 2>&1; grep "cases" $O/fuzz_grad_orders_93.txt
FUZZ_WIDE=1 FUZZ_ORDER=1 timeout 2400 python tools/fuzz_grad.py 400 94 > $O/fuzz_grad_wide_orders_94.txt 2>&1; grep "cases" $O/fuzz_grad_wide_orders_94.txt
