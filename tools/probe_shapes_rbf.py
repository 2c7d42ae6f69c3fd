"""SignatureRBF, order 1: covariances (Kzz, Kzx, Kxx-diag; T = 512 tensors with increments, N = 2,048 sequences) and a sequence Gram (N = 384), forward and forward + backward,
one shape parameter varied at a time around L = 50, d = 6, num_levels = 4 -- a search for cliffs between the instance families.  python tools/probe_shapes_rbf.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import autodiff, kernels
T, N, NG = 512, 2048, 384
rng = np.random.default_rng(0)
def timed(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
cases = [("base", 50, 6, 4)] + [("L", L, 6, 4) for L in (8, 20, 65, 100, 129, 200, 300)] + [("d", 50, d, 4) for d in (1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 24, 33)] + [("M", 50, 6, M) for M in (1, 2, 3, 5, 6, 7, 8)]
for what, L, d, M in cases:
    try:
        X = torch.as_tensor(np.cumsum(rng.standard_normal((N, L, d)) * 0.2, axis=1).reshape(N, -1), device="cuda:0")
        mod = autodiff.SignatureKernelModule(kernels.SignatureRBF(L * d, d, M, lengthscales=np.sqrt(d)), device="cuda:0")
        Z = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, 2, d)) * 0.4, device="cuda:0").requires_grad_(True)
        def covs_f():
            with torch.no_grad(): return mod.K_tens_n_seq_covs(Z, X, increments=True)
        def covs_fb():
            Z.grad = None; mod.zero_grad(set_to_none=True)
            a, b, c = mod.K_tens_n_seq_covs(Z, X, increments=True); (a.sum() + (b * b).sum() + c.sum()).backward()
        Xg = X[:NG].clone().requires_grad_(True)
        def gram_f():
            with torch.no_grad(): return mod.K(Xg)
        def gram_fb():
            Xg.grad = None; mod.zero_grad(set_to_none=True); o = mod.K(Xg); (o * o).sum().backward()
        work = T * N * L * (M * (M + 1) // 2) * d / 1e9
        print("%-4s L=%3d d=%2d M=%d | covariances f %7.2f f+b %8.2f ms (%.2f ms per Gcol-step) | Gram f %7.2f f+b %8.2f ms"
              % (what, L, d, M, timed(covs_f), timed(covs_fb), 0.0 if work == 0 else 1.0, timed(gram_f), timed(gram_fb)), flush=True)
    except Exception as e:
        print("%-4s L=%3d d=%2d M=%d | FAILED %s: %s" % (what, L, d, M, type(e).__name__, str(e)[:140]), flush=True)
