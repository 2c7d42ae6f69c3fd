"""K(X) at BASELINE configs[1]'s size with SignatureCosine: through the feature contraction (the cosine kernel is the linear kernel of the
unit vectors x / |x|) and through the pair kernels."""
import os
import sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import kernels, _lib
N, L, d, M = 4096, 64, 8, 5
X = torch.as_tensor(1.0 + np.random.default_rng(0).standard_normal((N, L * d)), device="cuda:0")
ctx = _lib.context(0, torch.cuda.current_stream(torch.device("cuda:0")).cuda_stream)
for route in (-1, 0):
    ctx.set_option("sig_features", route)
    k = kernels.SignatureCosine(L * d, d, M)
    k.K(X); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): k.K(X)
    torch.cuda.synchronize()
    print("cosine, sig_features", route, (time.perf_counter() - t0) / 3 * 1e3, "ms")
ctx.set_option("sig_features", -1)
