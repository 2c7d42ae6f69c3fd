"""BASELINE.json configs[4]: RBF signature kernel, N=2048, L=128, d=16, num_levels=6, fp32 (and the same in fp64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels, _lib
N, L, d, M = 2048, 128, 16, 6
rng = np.random.default_rng(0)
X64 = np.cumsum(0.1 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1)
ctx = _lib.context(0, torch.cuda.current_stream().cuda_stream)
for base in ("rbf", "linear"):
    for dt in (torch.float32, torch.float64):
        X = torch.as_tensor(X64, device="cuda:0").to(dt)
        kern = (kernels.SignatureRBF if base == "rbf" else kernels.SignatureLinear)(L * d, d, M, lengthscales=d ** 0.5)
        kern.K(X); torch.cuda.synchronize()
        ctx.timing_reset(); t0 = time.perf_counter()
        for _ in range(3): out = kern.K(X)
        torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 3
        ms, n, pairs = ctx.timing_get()
        bp = 2 * L * d * X.element_size() + X.element_size()
        print(f"C5 {base} {dt}: {dtm*1e3:.2f} ms per K(X), pair kernel {ms/3:.2f} ms; pairs/s {N*N/dtm:.3e}; stream frac {N*N*bp/dtm/8e12:.3f}; diag {float(out[3,3]):.6f}")
