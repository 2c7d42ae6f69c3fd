#!/usr/bin/env python3
"""Where the time of one low-rank draw goes (landmarks, whitening, sketches) at BASELINE configs[2]'s shape: cProfile of draw_low_rank."""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gpsig_amd import kernels
N, L, d, M, T = 16384, 50, 6, 4, 512
rng = np.random.default_rng(0)
X = torch.as_tensor(np.cumsum(0.2 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1), device="cuda")
Z = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, d)), device="cuda")
k = kernels.SignatureRBF(L * d, d, M, lengthscales=float(np.sqrt(d)), low_rank=True)
k.draw_low_rank(X=X, Z=Z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    k.draw_low_rank(X=X, Z=Z)
print("draw_low_rank: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    k.draw_low_rank(X=X, Z=Z)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
