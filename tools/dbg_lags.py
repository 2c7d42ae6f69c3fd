import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import autodiff, kernels
N, T, L, d, M = 2048, 512, 50, 4, 4
dev = torch.device("cuda:0"); rng = np.random.default_rng(0)
def tm(f, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for lags in (0, 1, 2):
    kern = kernels.SignatureRBF(L * d, d, M, lengthscales=2.0, num_lags=lags or None)
    mod = autodiff.SignatureKernelModule(kern, device=dev)
    de = d * (lags + 1)
    X = torch.tensor(np.cumsum(0.2 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1), device=dev, requires_grad=True)
    Z = torch.tensor(rng.standard_normal((M * (M + 1) // 2, T, 2, de)), device=dev, requires_grad=True)
    fwd = lambda: sum(a.sum() for a in mod.K_tens_n_seq_covs(Z, X, increments=True))
    with torch.no_grad():
        a = [tm(fwd, 1) for _ in range(6)]
    b = [tm(fwd, 1) for _ in range(4)]
    with torch.no_grad():
        c = tm(lambda: mod.scale_sequences(mod._seq3(X)))
        Xs = mod.scale_sequences(mod._seq3(X)); Zs = mod.scale_tensors(Z)
        e = tm(lambda: mod._diag_levels(Xs)); f = tm(lambda: mod._tens_levels(Zs, True))
        fac = torch.ones((M + 1, N), dtype=torch.float64, device=dev)
        g = tm(lambda: mod._tvs_weighted(Zs, Xs, fac, True))
    print(f"lags={lags}: no_grad forwards {np.round(a, 2)}  grad-mode forwards {np.round(b, 2)}  scale {c:.2f} diag {e:.2f} kzz {f:.2f} kzx {g:.2f}", flush=True)
