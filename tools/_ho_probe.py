import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import kernels, autodiff
T, N, L, d, M = 512, 16384, 50, 6, 4
rng = np.random.default_rng(0)
X = torch.as_tensor(rng.standard_normal((N, L * d)) * 0.3, device="cuda:0")
Z = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, d)) * 0.3, device="cuda:0")
order = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mod = autodiff.SignatureKernelModule(kernels.SignatureRBF(L * d, d, M, order=order), device="cuda:0")
Zp = Z.clone().requires_grad_(True)
for _ in range(3):
    Zp.grad = None
    mod.zero_grad(set_to_none=True)
    o = mod.K_tens_vs_seq(Zp, X, increments=False)
    (o * o).sum().backward()
torch.cuda.synchronize()
