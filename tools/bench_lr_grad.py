"""Low-rank mode, forward + backward of the SVGP covariances through SignatureKernelModule (torch-op route of autodiff.py) against the exact
mode at the same shapes:  python tools/bench_lr_grad.py [N] [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import autodiff, kernels

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
L, d, M = 50, 6, 4
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
X = torch.tensor(np.cumsum(0.2 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1), device=dev)
Z = torch.tensor(rng.standard_normal((M * (M + 1) // 2, T, d)), device=dev, requires_grad=True)
for low_rank in (False, True):
    kern = kernels.SignatureRBF(L * d, d, M, lengthscales=d ** 0.5, low_rank=low_rank, num_components=50, rank_bound=50)
    kern.rng = np.random.default_rng(1)
    mod = autodiff.SignatureKernelModule(kern, device=dev)
    W = torch.tensor(rng.standard_normal((T, N)), device=dev)
    def fwd():
        Kzz, Kzx, Kxx = mod.K_tens_n_seq_covs(Z, X)
        return Kzz.sum() + (Kzx * W).sum() + Kxx.sum()
    def step():
        mod.zero_grad(); Z.grad = None
        fwd().backward()
    for f, name in ((lambda: fwd(), "forward"), (step, "forward+backward")):
        for _ in range(2): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize()
        print(f"T={T} N={N} L={L} d={d} M={M} low_rank={low_rank}: {name} {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)
