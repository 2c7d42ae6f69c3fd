"""Gradient-path timings on one MI355X: (a) the SVGP training step of the reference's ts_classification notebook
(5000 iterations in 109 s = 46 it/s with the kernel fixed, 2.24 s per 100 iterations with the kernel trainable,
on an unnamed CUDA GPU), (b) forward + backward of the three SVGP covariances at BASELINE configs[2], (c) forward + backward
of a full Gram."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels, models, likelihoods as LK, inducing_variables as iv, autodiff

dev = torch.device("cuda:0")
for _env, _opt in (("GPSIG_TVS_ZREG", "tvs_zreg"), ("GPSIG_GRAD_IMPL", "grad_impl"), ("GPSIG_TVS_GRAD_TILE", "tvs_grad_tile")):
    if os.environ.get(_env):
        from gpsig_amd import _lib
        _lib.context(0, torch.cuda.current_stream(dev).cuda_stream).set_option(_opt, int(os.environ[_env]))
rng = np.random.default_rng(0)


def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n


which = sys.argv[1:] or ["a", "d", "b", "c"]
if "a" in which:
    # notebooks/ts_classification.ipynb: LIBRAS, N_train=144 (here synthetic of that shape), L=45, d=3 (+ time? no), M=4, 200 inducing tensors
    # with increments, minibatch 50, 15 classes, SignatureRBF
    N, L, d, M, T, C, mb = 144, 45, 3, 4, 200, 15, 50
    X = torch.tensor(np.cumsum(rng.standard_normal((N, L, d)) * 0.2, axis=1).reshape(N, -1), device=dev)
    Y = torch.tensor(rng.integers(0, C, (N, 1)).astype(np.float64), device=dev)
    Z = 0.3 * rng.standard_normal((M * (M + 1) // 2, T, 2, d))
    kern = kernels.SignatureRBF(L * d, d, M, lengthscales=1.0)
    model = models.SVGPModule(kern, iv.InducingTensors(Z, M, increments=True), LK.MultiClass(C), num_latent=C, num_data=N, device=dev)
    for trainable in (False, True):
        for p in model.kernel.parameters(): p.requires_grad_(trainable)
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
        idx = torch.arange(mb, device=dev)
        def step():
            opt.zero_grad(); loss = -model.elbo(X[idx], Y[idx]); loss.backward(); opt.step()
        dt = timeit(step, n=50, warm=5)
        print(f"(a) SVGP training step (LIBRAS shape, minibatch {mb}, {T} inducing tensors, kernel trainable={trainable}): {dt*1e3:.2f} ms = {1/dt:.1f} it/s")
if "d" in which:
    # benchmarks/run_gpsig_benchmarks.py:32: num_levels=4, 500 inducing tensors with increments, num_lags=1, minibatch 50; a mid-sized data set
    N, L, d, M, T, C, mb = 400, 200, 3, 4, 500, 5, 50
    Xn = np.cumsum(rng.standard_normal((N, L, d)) * 0.1, axis=1)
    X = torch.tensor(Xn.reshape(N, -1), device=dev)
    Y = torch.tensor(rng.integers(0, C, (N, 1)).astype(np.float64), device=dev)
    from gpsig_amd import utils
    Z = utils.suggest_initial_inducing_tensors(Xn, M, T, increments=True, num_lags=1, rng=rng)
    kern = kernels.SignatureRBF(L * d, d, M, lengthscales=1.0, num_lags=1)
    model = models.SVGPModule(kern, iv.InducingTensors(Z, M, increments=True), LK.MultiClass(C), num_latent=C, num_data=N, device=dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    idx = torch.arange(mb, device=dev)
    def step():
        opt.zero_grad(); loss = -model.elbo(X[idx], Y[idx]); loss.backward(); opt.step()
    dt = timeit(step, n=20, warm=3)
    print(f"(d) SVGP training step (benchmark settings: 500 inducing tensors, increments, num_lags=1, minibatch {mb}, L={L}, d={d}): {dt*1e3:.2f} ms = {1/dt:.1f} it/s")
if "b" in which:
    T, N, L, d, M = 512, 16384, 50, 6, 4
    X = torch.tensor(np.cumsum(0.2 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1), device=dev)
    for base in (os.environ.get("BENCH_GRAD_BASES") or "rbf,linear").split(","):
        kern = (kernels.SignatureRBF if base == "rbf" else kernels.SignatureLinear)(L * d, d, M, lengthscales=d ** 0.5)
        mod = autodiff.SignatureKernelModule(kern, device=dev)
        mod.feature_route = os.environ.get("GPSIG_FEATURE_ROUTE", "1") != "0"     # 0: linear Kzx / diagonals through the recursions' kernels
        for incr in [bool(int(v)) for v in (os.environ.get("BENCH_GRAD_INCR") or "0,1").split(",")]:
            Z = torch.tensor(rng.standard_normal((M * (M + 1) // 2, T, 2, d) if incr else (M * (M + 1) // 2, T, d)), device=dev, requires_grad=True)
            W = torch.tensor(rng.standard_normal((T, N)), device=dev)
            def fwd():
                with torch.no_grad(): mod.K_tens_n_seq_covs(Z, X, increments=incr)
            def both():
                mod.zero_grad(); Z.grad = None
                Kzz, Kzx, Kxx = mod.K_tens_n_seq_covs(Z, X, increments=incr)
                ((Kzx * W).sum() + Kzz.sum() + Kxx.sum()).backward()
            tf, tb = timeit(fwd, 3, 1), timeit(both, 3, 1)
            print(f"(b) C3 {base} incr={incr}{'' if mod.feature_route or base != 'linear' else ' [recursion kernels]'}: forward {tf*1e3:.1f} ms, forward+backward {tb*1e3:.1f} ms")
if "c" in which:
    for (N, L, d, M) in ((512, 64, 8, 5), (1024, 64, 8, 5)):
        X = torch.tensor(rng.standard_normal((N, L * d)), device=dev)
        for base in ("linear", "rbf"):
            kern = (kernels.SignatureRBF if base == "rbf" else kernels.SignatureLinear)(L * d, d, M, lengthscales=(d ** 0.5 if base == "rbf" else 1.0))
            mod = autodiff.SignatureKernelModule(kern, device=dev)
            W = torch.tensor(rng.standard_normal((N, N)), device=dev)
            def fwd():
                with torch.no_grad(): mod.K(X)
            def both():
                mod.zero_grad(); (mod.K(X) * W).sum().backward()
            tf, tb = timeit(fwd, 3, 1), timeit(both, 3, 1)
            print(f"(c) full Gram N={N} L={L} d={d} M={M} {base}: forward {tf*1e3:.1f} ms, forward+backward {tb*1e3:.1f} ms ({N*N/tb:.3e} pairs/s)")
