"""Kzx at BASELINE configs[2]'s size (T = 512 inducing tensors, N = 16,384, L = 50, d = 6, num_levels = 4) for the first- and
higher-order algorithms (signature_algs.py:101-160), SignatureLinear and SignatureRBF."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import kernels  # noqa: E402

T, N, L, d, M = 512, 16384, 50, 6, 4
rng = np.random.default_rng(0)
X = torch.as_tensor(rng.standard_normal((N, L * d)) * 0.3, device="cuda:0")
Z = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, d)) * 0.3, device="cuda:0")
for cls in (kernels.SignatureLinear, kernels.SignatureRBF):
    for order in (1, 2, 4):
        k = cls(L * d, d, M, order=order)
        k.K_tens_vs_seq(Z, X); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            k.K_tens_vs_seq(Z, X)
        torch.cuda.synchronize()
        print(cls.__name__, "order", order, "%.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
