#!/bin/bash
# round 6: randomised sweeps with the round's library -- the usual distribution, and the widths of the reference's run settings (FUZZ_WIDE=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06k; mkdir -p $O
timeout 1200 python tools/fuzz_parity.py 500 81 > $O/fuzz_parity_81.txt 2>&1; tail -3 $O/fuzz_parity_81.txt | cut -c1-300
FUZZ_WIDE=1 timeout 1500 python tools/fuzz_parity.py 300 82 > $O/fuzz_parity_wide_82.txt 2>&1; tail -3 $O/fuzz_parity_wide_82.txt | cut -c1-300
timeout 1200 python tools/fuzz_grad.py 300 83 > $O/fuzz_grad_83.txt 2>&1; tail -3 $O/fuzz_grad_83.txt | cut -c1-300
FUZZ_WIDE=1 timeout 1500 python tools/fuzz_grad.py 200 84 > $O/fuzz_grad_wide_84.txt 2>&1; tail -3 $O/fuzz_grad_wide_84.txt | cut -c1-300
FUZZ_ORDER=1 timeout 900 python tools/fuzz_grad.py 100 85 > $O/fuzz_grad_orders_85.txt 2>&1; tail -2 $O/fuzz_grad_orders_85.txt | cut -c1-300
