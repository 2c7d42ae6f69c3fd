#!/bin/bash
# round 5: the Matern families at compile time in the sequence Gram's evaluation kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05aa; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python - > $O/matern_forward.txt 2>&1 <<'PY'
import math, time, sys
import numpy as np, torch
sys.path.insert(0, '.')
from gpsig_amd import _lib, autodiff, kernels
dev = torch.device('cuda:0')
ctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
N, L, D, M = 4096, 64, 8, 5
rng = np.random.default_rng(0)
X = torch.tensor(np.cumsum(rng.standard_normal((N, L, D)) * 0.3, 1).reshape(N, -1), device=dev)
for cls in (kernels.SignatureMatern12, kernels.SignatureMatern32, kernels.SignatureMatern52, kernels.SignatureRBF):
    kern = cls(L * D, D, M, lengthscales=math.sqrt(D))
    res = {}
    for fast in (1, 0):
        ctx.set_option('matern_fast', fast)
        K = kern.K(X); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): K = kern.K(X)
        torch.cuda.synchronize()
        res[fast] = ((time.perf_counter() - t0) / 3 * 1e3, K)
    ctx.set_option('matern_fast', 1)
    err = float((res[1][1] - res[0][1]).abs().max() / res[0][1].abs().max())
    print(f"{cls.__name__} K(X) N={N}: compile-time instance {res[1][0]:.1f} ms, run-time kind {res[0][0]:.1f} ms, rel. difference {err:.2e}", flush=True)
N = 1024
X = torch.tensor(np.random.default_rng(0).standard_normal((N, L * D)), device=dev)
W = torch.tensor(np.random.default_rng(1).standard_normal((N, N)), device=dev)
for cls in (kernels.SignatureMatern32, kernels.SignatureMatern12):
    mod = autodiff.SignatureKernelModule(cls(L * D, D, M, lengthscales=math.sqrt(D)), device=dev)
    def step():
        mod.zero_grad(); (mod.K(X) * W).sum().backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    print(cls.__name__, 'K(X) forward + backward, 1,024 sequences:', round((time.perf_counter() - t0) / 5 * 1e3, 2), 'ms', flush=True)
PY
cat $O/matern_forward.txt
