import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import kernels, autodiff
from oracle import sigkern_oracle as O
for d, L, N in ((28, 40, 20), (70, 30, 12), (126, 50, 10)):
    for order in (2, 3):
        M = 4
        X = np.cumsum(np.random.default_rng(0).standard_normal((N, L, d)) * 0.3 / np.sqrt(d), axis=1).reshape(N, -1)
        try:
            k = kernels.SignatureRBF(L * d, d, M, order=order, lengthscales=1.0)
            ko = O.SignatureKernelOracle(L * d, d, M, base="rbf", order=order, lengthscales=1.0)
            Xt = torch.tensor(X, device="cuda:0")
            t0 = time.perf_counter(); K = k.K(Xt); Kd = k.Kdiag(Xt); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            want = ko.K(X)
            print(d, order, "K ok", float(np.abs(K.cpu().numpy() - want).max() / np.abs(want).max()), "%.1f ms" % (dt * 1e3), flush=True)
            mod = autodiff.SignatureKernelModule(k, device="cuda:0")
            Xg = Xt.clone().requires_grad_(True)
            (mod.K(Xg) ** 2).sum().backward()
            print(d, order, "grad ok", bool(torch.isfinite(Xg.grad).all()), flush=True)
        except Exception as e:
            print(d, order, "FAILED", type(e).__name__, str(e)[:200], flush=True)
