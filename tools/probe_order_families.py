"""Kzx forward and forward + backward at BASELINE configs[2]'s size for orders 1 / 2 and the base-kernel families beyond SignatureRBF (which tools/bench_order_kzx.py covers)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import autodiff, kernels
T, N, L, d, M = 512, int(os.environ.get("PROBE_N", 16384)), 50, 6, 4
rng = np.random.default_rng(0)
X = torch.as_tensor(rng.standard_normal((N, L * d)) * 0.3, device="cuda:0")
Z = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, d)) * 0.3, device="cuda:0")
for cls in (kernels.SignatureRBF, kernels.SignatureMatern32, kernels.SignaturePoly, kernels.SignatureLinear):
    for order in (1, 2):
        mod = autodiff.SignatureKernelModule(cls(L * d, d, M, order=order), device="cuda:0")
        Zp = Z.clone().requires_grad_(True)
        def fwd():
            with torch.no_grad():
                return mod.K_tens_vs_seq(Zp, X)
        def fb():
            Zp.grad = None; mod.zero_grad(set_to_none=True)
            o = mod.K_tens_vs_seq(Zp, X); (o * o).sum().backward()
        ts = []
        for fn in (fwd, fb):
            fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(2): fn()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 2 * 1e3)
        print("%-18s order %d: Kzx forward %8.2f ms, forward + backward %9.2f ms" % (cls.__name__, order, ts[0], ts[1]), flush=True)
