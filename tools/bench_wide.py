"""K(X), N = 4,096 sequences of L = 64 points, SignatureLinear on wider state spaces: feature contraction against the pair recursion."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import _lib, kernels  # noqa: E402

N, L = 4096, 64
ctx = _lib.context(0, torch.cuda.current_stream(torch.device("cuda:0")).cuda_stream)
for d, M in ((16, 3), (12, 4), (16, 2), (10, 4), (32, 3), (24, 3), (32, 2)):
    X = torch.as_tensor(np.random.default_rng(0).standard_normal((N, L * d)) / np.sqrt(d), device="cuda:0")
    k = kernels.SignatureLinear(L * d, d, M)
    res = {}
    for route in (-1, 0):
        ctx.set_option("sig_features", route)
        G = k.K(X); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            G = k.K(X)
        torch.cuda.synchronize()
        res[route] = ((time.perf_counter() - t0) / 3 * 1e3, G)
    ctx.set_option("sig_features", -1)
    err = float((res[-1][1] - res[0][1]).abs().max() / res[0][1].abs().max())
    print(f"d={d} M={M}: contraction {res[-1][0]:.2f} ms, pair recursion {res[0][0]:.2f} ms, max rel. difference {err:.1e}")
