import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from gpsig_amd import kernels
rng = np.random.default_rng(0)
for N in (256, 1024):
    L, d, M = 64, 8, 5
    X = torch.as_tensor(np.cumsum(0.2 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1), device="cuda")
    for name, k in (("spectral", kernels.SignatureSpectral(L * d, d, M, Q=5)), ("rbf", kernels.SignatureRBF(L * d, d, M))):
        k.K(X); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): k.K(X)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(name, N, "K(X) %.2f ms" % (dt * 1e3), "%.3e pairs/s" % (N * N / dt))
