#!/bin/bash
# bench.py --gpus 2 end to end on ONE GPU: two ranks over gloo on the same device (functional check of the N > 1 path:
# BASELINE configs[3], N=32768, owned-row blocks, chunked gather, symmetrisation, verify_max_abs_diff_vs_single_rank).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r02
GPSIG_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r02/bench_c4_2ranks_gloo.json 2> gpurun_out/r02/bench_c4_2ranks_gloo.err
tail -1 gpurun_out/r02/bench_c4_2ranks_gloo.json | cut -c1-1500
tail -3 gpurun_out/r02/bench_c4_2ranks_gloo.err
# the sequence-sharded SVGP covariances (BASELINE configs[2]) the same way
GPSIG_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 \
    bench.py --gpus 2 --config c3 --steps 3 --warmup 1 > gpurun_out/r02/bench_c3_2ranks_gloo.json 2> gpurun_out/r02/bench_c3_2ranks_gloo.err
tail -1 gpurun_out/r02/bench_c3_2ranks_gloo.json | cut -c1-400
tail -3 gpurun_out/r02/bench_c3_2ranks_gloo.err
