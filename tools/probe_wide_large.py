"""Kzx at 12 columns (d = 6, num_lags = 1: the reference's lag setting) for MANY sequences (prediction-sized: N = 4,096, T = 512 tensors with increments, L = 50, num_levels = 4):
the wide route away from the minibatch regime.  python tools/probe_wide_large.py [fwd|fb]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import autodiff, kernels
N, T, L, d, M = 4096, 512, 50, 6, 4
rng = np.random.default_rng(0)
X = torch.as_tensor(np.cumsum(rng.standard_normal((N, L, d)) * 0.2, axis=1).reshape(N, -1), device="cuda:0")
mod = autodiff.SignatureKernelModule(kernels.SignatureRBF(L * d, d, M, num_lags=1, lengthscales=np.sqrt(d)), device="cuda:0")
Z = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, 2, 2 * d)) * 0.4, device="cuda:0").requires_grad_(True)
what = sys.argv[1] if len(sys.argv) > 1 else "fb"
for _ in range(3):
    if what == "fwd":
        with torch.no_grad():
            mod.K_tens_vs_seq(Z, X, increments=True)
    else:
        Z.grad = None; mod.zero_grad(set_to_none=True)
        o = mod.K_tens_vs_seq(Z, X, increments=True); (o * o).sum().backward()
torch.cuda.synchronize()
