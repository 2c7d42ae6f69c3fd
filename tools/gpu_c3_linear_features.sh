#!/bin/bash
# BASELINE configs[2] with SignatureLinear, forward + backward: tensor-vs-sequence recursion kernels against the level-feature route
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
BENCH_GRAD_BASES=linear GPSIG_FEATURE_ROUTE=0 python tools/bench_grad.py b 2>&1 | grep "(b)"
BENCH_GRAD_BASES=linear python tools/bench_grad.py b 2>&1 | grep "(b)"
BENCH_GRAD_BASES=rbf python tools/bench_grad.py b 2>&1 | grep "(b)"
python tools/bench_matmul_shapes.py 2>&1 | grep -v amdgpu
} | tee gpurun_out/bench_grad_c3_linear_features.txt
tools/gpu_prof_c3lin.sh > /dev/null 2>&1
