#!/bin/bash
# round 5: the fused reverse kernel with 32 lanes per pair (column sides of 65 .. 128 points, two pairs per wavefront)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05q; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_grad.py -q -x -k 'stationary_kernels or route_taken or wave_and_storage or seq_level or beyond_64 or wide_state or inducing_sequences' > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for impl in 0 4; do
  GPSIG_GRAD_IMPL=$impl timeout 300 python tools/bench_grad_gram.py 512 rbf 3 128 8 5 2>&1 | tail -1 | sed "s/^/grad_impl=$impl /" >> $O/grad_g32.txt
  GPSIG_GRAD_IMPL=$impl timeout 300 python tools/bench_grad_gram.py 512 rbf 3 100 4 4 2>&1 | tail -1 | sed "s/^/grad_impl=$impl /" >> $O/grad_g32.txt
  GPSIG_GRAD_IMPL=$impl timeout 300 python tools/bench_grad_gram.py 512 rbf 3 64 12 4 2>&1 | tail -1 | sed "s/^/grad_impl=$impl /" >> $O/grad_g32.txt
done
cat $O/grad_g32.txt
