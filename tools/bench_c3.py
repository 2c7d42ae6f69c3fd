"""BASELINE.json configs[2]: SVGP inducing-tensor path, Kzz + Kzx + Kxx-diag, T=512, N=16384, L=50, d=6, num_levels=4, RBF."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels, _lib
T, N, L, d, M = 512, 16384, 50, 6, 4
rng = np.random.default_rng(0)
X = torch.as_tensor(np.cumsum(0.2 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1), device="cuda:0")
ctx = _lib.context(0, torch.cuda.current_stream().cuda_stream)
for base in ("rbf", "linear"):
    for incr in (False, True):
        Z = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, 2, d) if incr else (M * (M + 1) // 2, T, d)), device="cuda:0")
        kern = (kernels.SignatureRBF if base == "rbf" else kernels.SignatureLinear)(L * d, d, M, lengthscales=d ** 0.5)
        kern.K_tens_n_seq_covs(Z, X, increments=incr); torch.cuda.synchronize()
        ctx.timing_reset(); t0 = time.perf_counter()
        for _ in range(3): kern.K_tens_n_seq_covs(Z, X, increments=incr)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        ms, n, pairs = ctx.timing_get()
        bp = L * d * 8 + (2 if incr else 1) * (M * (M + 1) // 2) * d * 8 + 8
        print(f"C3 {base} incr={incr}: {dt*1e3:.2f} ms per call (Kzz+Kzx+Kxx-diag), timed kernels {ms/3:.2f} ms; Kzx pairs/s {T*N/dt:.3e}; stream frac {T*N*bp/dt/8e12:.3f}")
