"""K(X) at BASELINE configs[1]'s size (N = 4,096, L = 64, d = 8, num_levels = 5, SignatureLinear) for the first- and higher-order algorithms
(signature_algs.py:8-74): through the feature contraction, and -- `--lattice` -- through the pair kernels."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import _lib, kernels
if "--lattice" in sys.argv:
    _lib.context(0, torch.cuda.current_stream(torch.device("cuda:0")).cuda_stream).set_option("sig_features", 0)
N, L, d, M = 4096, 64, 8, 5
X = torch.as_tensor(np.random.default_rng(0).standard_normal((N, L * d)), device="cuda:0")
for order in (1, 2, 3, 5):
    k = kernels.SignatureLinear(L * d, d, M, order=order)
    k.K(X); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): k.K(X)
    torch.cuda.synchronize()
    print("order", order, (time.perf_counter() - t0) / 3 * 1e3, "ms")
