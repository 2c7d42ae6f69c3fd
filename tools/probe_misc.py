"""Corners beside the main path, SignatureRBF unless named: inducing SEQUENCES (K_seq_n_seq_covs), normalization / difference off, the spectral kernel, low-rank mode -- forward and
forward + backward at T = 256, N = 2,048, L = 50, d = 6, num_levels = 4.  python tools/probe_misc.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import autodiff, kernels
T, N, L, d, M, Lz = 256, 2048, 50, 6, 4, 10
rng = np.random.default_rng(0)
X = torch.as_tensor(np.cumsum(rng.standard_normal((N, L, d)) * 0.2, axis=1).reshape(N, -1), device="cuda:0")
def timed(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def run(name, kern, seqs=False, **kw):
    try:
        mod = autodiff.SignatureKernelModule(kern, device="cuda:0")
        if seqs:
            Z = torch.as_tensor(np.cumsum(rng.standard_normal((T, Lz, d)) * 0.3, axis=1).reshape(T, -1), device="cuda:0").requires_grad_(True)
            call = lambda: mod.K_seq_n_seq_covs(Z, X)
        else:
            Z = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, 2, d)) * 0.4, device="cuda:0").requires_grad_(True)
            call = lambda: mod.K_tens_n_seq_covs(Z, X, increments=True)
        def f():
            with torch.no_grad(): return call()
        def fb():
            Z.grad = None; mod.zero_grad(set_to_none=True)
            a, b, c = call(); (a.sum() + (b * b).sum() + c.sum()).backward()
        print("%-46s covariances f %8.2f  f+b %9.2f ms" % (name, timed(f), timed(fb)), flush=True)
    except Exception as e:
        print("%-46s FAILED %s: %s" % (name, type(e).__name__, str(e)[:150]), flush=True)
ls = np.sqrt(d)
run("inducing tensors (reference point)", kernels.SignatureRBF(L * d, d, M, lengthscales=ls))
run("inducing sequences of 10 observations", kernels.SignatureRBF(L * d, d, M, lengthscales=ls), seqs=True)
run("inducing sequences, order 2", kernels.SignatureRBF(L * d, d, M, lengthscales=ls, order=2), seqs=True)
run("normalization off", kernels.SignatureRBF(L * d, d, M, lengthscales=ls, normalization=False))
run("difference off", kernels.SignatureRBF(L * d, d, M, lengthscales=ls, difference=False))
run("lengthscales None", kernels.SignatureRBF(L * d, d, M, lengthscales=None))
run("low-rank mode (50 components, rank 50)", kernels.SignatureRBF(L * d, d, M, lengthscales=ls, low_rank=True, num_components=50, rank_bound=50))
try:
    run("SignatureSpectral (3 mixture components)", kernels.SignatureSpectral(L * d, d, M, num_mixtures=3))
except Exception as e:
    print("spectral FAILED to construct:", type(e).__name__, str(e)[:150])
run("SignatureMatern32, inducing sequences", kernels.SignatureMatern32(L * d, d, M, lengthscales=ls), seqs=True)
run("SignatureLinear, inducing sequences", kernels.SignatureLinear(L * d, d, M, lengthscales=ls), seqs=True)
