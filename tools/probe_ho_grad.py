"""K(X) forward + backward at order 2 (SignatureRBF, N = 512, L = 64, d = 8, num_levels = 4), three repetitions: the workload of bench.py's
`grad-n512-rbf-order2` record on its own, for kernel traces and counter passes (tools/gpu_r6_m.sh)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import autodiff, kernels  # noqa: E402

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
