#!/bin/bash
# round 5: the fused reverse kernel without differences
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05p; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_grad.py -q -x -k 'stationary_kernels or route_taken or wave_and_storage or seq_level or beyond_64 or wide_state or inducing_sequences or module_gradients' > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python tools/fuzz_grad.py 250 93 > $O/fuzz_grad_93.txt 2>&1; tail -3 $O/fuzz_grad_93.txt | cut -c1-300
