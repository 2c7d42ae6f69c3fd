"""Forward and forward + backward of the SVGP covariances through gpsig_amd.autodiff for every training route (base kernel families, orders, lags,
inducing tensors / sequences, low-rank, the matrix route), at a minibatch shape and at a larger one: a sweep for outliers.
    python tools/bench_train_paths.py [small|large]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import autodiff, kernels

which = sys.argv[1] if len(sys.argv) > 1 else "small"
N, T, L, d, M = (50, 200, 50, 4, 4) if which == "small" else (2048, 512, 50, 4, 4)
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def tm(f, n=3):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


CASES = [
    ("rbf", dict(), "tensors"), ("rbf", dict(), "tensors-incr"), ("linear", dict(), "tensors"), ("matern32", dict(), "tensors"), ("poly", dict(), "tensors"),
    ("rbf", dict(num_lags=1), "tensors-incr"), ("rbf", dict(order=2), "tensors"), ("rbf", dict(order=4), "tensors"), ("linear", dict(order=4), "tensors"),
    ("rbf", dict(normalization=False), "tensors"), ("rbf", dict(difference=False), "tensors"),
    ("rbf", dict(), "sequences"), ("linear", dict(), "sequences"), ("rbf", dict(order=2), "sequences"),
    ("rbf", dict(low_rank=True, num_components=50, rank_bound=50), "tensors"), ("rbf", dict(low_rank=True, num_components=50, rank_bound=50), "sequences"),
    ("spectral", dict(), "tensors"), ("rbf", dict(), "gram"), ("linear", dict(), "gram"), ("cosine", dict(), "gram"), ("linear", dict(order=3), "gram"),
]
CLS = {"rbf": kernels.SignatureRBF, "linear": kernels.SignatureLinear, "matern32": kernels.SignatureMatern32, "poly": kernels.SignaturePoly,
       "spectral": kernels.SignatureSpectral, "cosine": kernels.SignatureCosine}
for base, kw, what in CASES:
    try:
        extra = dict(family="exp", Q=3) if base == "spectral" else {}
        kern = CLS[base](L * d, d, M, lengthscales=(d ** 0.5 if base in ("rbf", "matern32") else 1.0), **kw, **extra)
        kern.rng = np.random.default_rng(1)
        mod = autodiff.SignatureKernelModule(kern, device=dev)
        de = d * ((kw.get("num_lags") or 0) + 1)
        X = torch.tensor(np.cumsum(0.2 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1), device=dev, requires_grad=True)
        lt = M * (M + 1) // 2
        if what.startswith("tensors"):
            inc = what.endswith("incr")
            Z = torch.tensor(rng.standard_normal((lt, T, 2, de) if inc else (lt, T, de)), device=dev, requires_grad=True)
            fwd = lambda: sum(a.sum() for a in mod.K_tens_n_seq_covs(Z, X, increments=inc))
        elif what == "sequences":
            Ls = 10
            Z = torch.tensor(np.cumsum(0.2 * rng.standard_normal((T, Ls, d)), axis=1).reshape(T, -1), device=dev, requires_grad=True)
            fwd = lambda: sum(a.sum() for a in mod.K_seq_n_seq_covs(Z, X))
        else:
            Z = None
            fwd = lambda: mod.K(X).sum()

        def step():
            mod.zero_grad(); X.grad = None
            if Z is not None: Z.grad = None
            fwd().backward()
        with torch.no_grad():
            f0 = tm(lambda: fwd())
        fb = tm(step)
        print(f"{which:5s} N={N} T={T} L={L} d={d} M={M}  {base:9s} {str(kw):58s} {what:13s} forward {f0:9.2f} ms   forward+backward {fb:9.2f} ms   x{fb / f0:6.1f}", flush=True)
    except Exception as e:
        print(f"{which:5s} {base:9s} {str(kw):58s} {what:13s} FAILED: {type(e).__name__}: {str(e)[:120]}", flush=True)
