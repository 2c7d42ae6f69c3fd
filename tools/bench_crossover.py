"""Where the planner's threshold sits: K(X, X2) of N1 x N2 sequences through the feature contraction (sig_features = 1) and through the
pair recursion (0), device time per evaluation."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import _lib, kernels  # noqa: E402

ctx = _lib.context(0, torch.cuda.current_stream(torch.device("cuda:0")).cuda_stream)
rng = np.random.default_rng(0)
for (L, d, M) in ((50, 6, 4), (64, 8, 5), (64, 16, 3)):
    k = kernels.SignatureLinear(L * d, d, M)
    for (n1, n2) in ((64, 64), (128, 128), (500, 50), (256, 256), (1000, 100), (1024, 1024)):
        X = torch.as_tensor(rng.standard_normal((n1, L * d)) / np.sqrt(d), device="cuda:0")
        Y = torch.as_tensor(rng.standard_normal((n2, L * d)) / np.sqrt(d), device="cuda:0")
        res = []
        for route in (1, 0):
            ctx.set_option("sig_features", route)
            for _ in range(3):
                k.K(X, Y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                k.K(X, Y)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 20 * 1e6)
        ctx.set_option("sig_features", -1)
        print(f"L={L} d={d} M={M}  {n1:5d} x {n2:5d} = {n1 * n2:8d} pairs: contraction {res[0]:8.1f} us, pair recursion {res[1]:8.1f} us")
