#!/bin/bash
# round 5: more randomised sweeps with the round's final library (stash route included)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05y; mkdir -p $O
timeout 1500 python tools/fuzz_grad.py 600 95 > $O/fuzz_grad_95.txt 2>&1; tail -3 $O/fuzz_grad_95.txt | cut -c1-300
FUZZ_ORDER=1 timeout 900 python tools/fuzz_grad.py 200 96 > $O/fuzz_grad_96_orders.txt 2>&1; tail -2 $O/fuzz_grad_96_orders.txt | cut -c1-300
timeout 1500 python tools/fuzz_parity.py 1000 75 > $O/fuzz_parity_75.txt 2>&1; tail -3 $O/fuzz_parity_75.txt | cut -c1-300
