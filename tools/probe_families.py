"""Forward and forward + backward times of the covariances an SVGP needs (Kzz, Kzx, Kxx-diag) and of a sequence Gram, for every base-kernel family, with and without
lags / increments: a search for slow corners (round 6 found the Matern families' Kzx reverse pass this way).  python tools/probe_families.py [N] [T]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import autodiff, kernels
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
L, d, M, NG = 50, 6, 4, 384
rng = np.random.default_rng(0)
X = torch.as_tensor(np.cumsum(rng.standard_normal((N, L, d)) * 0.2, axis=1).reshape(N, -1), device="cuda:0")
FAM = [("linear", kernels.SignatureLinear), ("cosine", kernels.SignatureCosine), ("poly", kernels.SignaturePoly), ("rbf", kernels.SignatureRBF), ("mix", kernels.SignatureMix),
       ("matern12", kernels.SignatureMatern12), ("matern32", kernels.SignatureMatern32), ("matern52", kernels.SignatureMatern52)]
def timed(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for name, cls in FAM:
    for lags, incr in ((0, False), (0, True), (1, True)):
        try:
            de = d * (lags + 1)
            kern = cls(L * d, d, M, num_lags=lags or None, lengthscales=np.sqrt(d), order=int(os.environ.get("PROBE_ORDER", "1")))
            mod = autodiff.SignatureKernelModule(kern, device="cuda:0")
            Z = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, 2, de) if incr else (M * (M + 1) // 2, T, de)) * 0.4, device="cuda:0").requires_grad_(True)
            def covs_f():
                with torch.no_grad(): return mod.K_tens_n_seq_covs(Z, X, increments=incr)
            def covs_fb():
                Z.grad = None; mod.zero_grad(set_to_none=True)
                a, b, c = mod.K_tens_n_seq_covs(Z, X, increments=incr); (a.sum() + (b * b).sum() + c.sum()).backward()
            Xg = X[:NG].clone().requires_grad_(True)
            def gram_f():
                with torch.no_grad(): return mod.K(Xg)
            def gram_fb():
                Xg.grad = None; mod.zero_grad(set_to_none=True); o = mod.K(Xg); (o * o).sum().backward()
            print("%-9s lags %d increments %d | covariances (T=%d, N=%d) f %7.2f  f+b %8.2f ms | Gram of %d f %7.2f  f+b %8.2f ms"
                  % (name, lags, incr, T, N, timed(covs_f), timed(covs_fb), NG, timed(gram_f), timed(gram_fb)), flush=True)
        except Exception as e:
            print("%-9s lags %d increments %d | FAILED %s: %s" % (name, lags, incr, type(e).__name__, str(e)[:120]), flush=True)
