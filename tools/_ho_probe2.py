import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import kernels, autodiff
N, L, d, M = 512, 64, 8, 4
order = int(sys.argv[1]) if len(sys.argv) > 1 else 2
X = torch.as_tensor(np.cumsum(np.random.default_rng(0).standard_normal((N, L, d)) * 0.3, axis=1).reshape(N, -1), device="cuda:0").requires_grad_(True)
mod = autodiff.SignatureKernelModule(kernels.SignatureRBF(L * d, d, M, order=order, lengthscales=np.sqrt(d)), device="cuda:0")
for _ in range(3):
    X.grad = None
    mod.zero_grad(set_to_none=True)
    o = mod.K(X)
    (o * o).sum().backward()
torch.cuda.synchronize()
