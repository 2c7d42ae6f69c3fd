"""Evaluation-path sweep for outliers: every kernel family x order x lags x dtype x (exact | low-rank) through the public classes
(CUDA tensors in, CUDA tensors out), K(X), K(X, X2), Kdiag, Kzz + Kzx + Kxx-diag, at one moderate size.
    python tools/bench_eval_paths.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels

N, N2, T, L, d, M = 2048, 512, 512, 50, 4, 4
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def tm(f, n=3):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


CLS = {"linear": kernels.SignatureLinear, "rbf": kernels.SignatureRBF, "cosine": kernels.SignatureCosine, "poly": kernels.SignaturePoly,
       "mix": kernels.SignatureMix, "matern12": kernels.SignatureMatern12, "matern32": kernels.SignatureMatern32, "matern52": kernels.SignatureMatern52,
       "spectral": kernels.SignatureSpectral}
rows = []
for base in CLS:
    for kw in (dict(), dict(order=2), dict(order=4), dict(num_lags=1), dict(normalization=False), dict(difference=False),
               dict(low_rank=True, num_components=50, rank_bound=50)):
        if base == "spectral" and kw.get("low_rank"):
            continue
        for dt in (torch.float64, torch.float32):
            if dt == torch.float32 and (kw and kw != dict(order=2)):
                continue
            try:
                extra = dict(family="exp", Q=3) if base == "spectral" else {}
                kern = CLS[base](L * d, d, M, lengthscales=(2.0 if base not in ("linear", "cosine", "poly") else 1.0), **kw, **extra)
                kern.rng = np.random.default_rng(1)
                de = d * ((kw.get("num_lags") or 0) + 1)
                X = torch.tensor(np.cumsum(0.2 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1), device=dev, dtype=dt)
                X2 = torch.tensor(np.cumsum(0.2 * rng.standard_normal((N2, L, d)), axis=1).reshape(N2, -1), device=dev, dtype=dt)
                Z = torch.tensor(rng.standard_normal((M * (M + 1) // 2, T, de)), device=dev, dtype=dt)
                t = [tm(lambda: kern.K(X)), tm(lambda: kern.K(X, X2)), tm(lambda: kern.Kdiag(X)), tm(lambda: kern.K_tens_n_seq_covs(Z, X))]
                print(f"{base:9s} {str(kw):60s} {'f32' if dt == torch.float32 else 'f64'}  K(X) {t[0]:9.2f}  K(X,X2) {t[1]:8.2f}  Kdiag {t[2]:7.2f}  covs {t[3]:8.2f} ms", flush=True)
            except Exception as e:
                print(f"{base:9s} {str(kw):60s} {'f32' if dt == torch.float32 else 'f64'}  FAILED {type(e).__name__}: {str(e)[:100]}", flush=True)
