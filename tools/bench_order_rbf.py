"""K(X) at BASELINE configs[1]'s shape (N = 4,096, L = 64, d = 8, num_levels = 5 and 4) with SignatureRBF for orders 1, 2, 4 (signature_algs.py:8-74):
the exact higher-order instances of round 6 against the run-time ones (option exact = 0), values compared with each other and -- a block of 24
sequences -- with the oracle.    python tools/bench_order_rbf.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import _lib, kernels  # noqa: E402
from oracle import sigkern_oracle as O  # noqa: E402

N, L, d = 4096, 64, 8
dev = torch.device("cuda:0")
ctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
X = torch.as_tensor(np.cumsum(np.random.default_rng(0).standard_normal((N, L, d)) * 0.3, axis=1).reshape(N, -1), device=dev)
for M in (5, 4):
    base = None
    for order in (1, 2, 4):
        k = kernels.SignatureRBF(L * d, d, M, order=order, lengthscales=np.sqrt(d))
        res = {}
        for exact in (1, 0):
            ctx.set_option("exact", exact)
            G = k.K(X); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                G = k.K(X)
            torch.cuda.synchronize()
            res[exact] = ((time.perf_counter() - t0) / 3 * 1e3, G)
        ctx.set_option("exact", 1)
        ko = O.SignatureKernelOracle(L * d, d, M, base="rbf", order=order, lengthscales=np.sqrt(d))
        want = ko.K(X[:24].cpu().numpy())
        err = float(np.abs(res[1][1][:24, :24].cpu().numpy() - want).max() / np.abs(want).max())
        diff = float((res[1][1] - res[0][1]).abs().max() / res[0][1].abs().max())
        if order == 1:
            base = res[1][0]
        print(f"num_levels={M} order={order}: exact instances {res[1][0]:8.2f} ms ({res[1][0] / base:5.2f} x order 1)   run-time instances {res[0][0]:8.2f} ms   "
              f"difference {diff:.1e}   vs oracle {err:.1e}", flush=True)

# the reverse pass (round 6: two sweeps of a wavefront per pair at order > 1, csrc/grad_wave_ho_kernel.hpp; option grad_impl = 1: the lattice operations it replaces)
from gpsig_amd import autodiff  # noqa: E402
print("\nforward and forward + backward of K(X) (autodiff module, gradient w.r.t. X, lengthscales, variances), N = 512, L = 64, d = 8, num_levels = 4:")
Ng, M = 512, 4
Xg = X[:Ng].clone().requires_grad_(True)
for order in (1, 2, 4):
    mod = autodiff.SignatureKernelModule(kernels.SignatureRBF(L * d, d, M, order=order, lengthscales=np.sqrt(d)), device="cuda:0")
    out = {}
    for impl in ((0, 1) if order > 1 else (0,)):
        ctx.set_option("grad_impl", impl)

        def fwd():
            with torch.no_grad():
                return mod.K(Xg)

        def fb():
            Xg.grad = None
            mod.zero_grad(set_to_none=True)
            o = mod.K(Xg)
            (o * o).sum().backward()
        ts = []
        for fn in (fwd, fb):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 3 * 1e3)
        out[impl] = (ts, Xg.grad.clone())
    ctx.set_option("grad_impl", 0)
    ts = out[0][0]
    extra = ""
    if 1 in out:
        extra = "   lattice operations: %.1f ms, gradients differ by %.1e" % (out[1][0][1], float((out[0][1] - out[1][1]).abs().max() / out[1][1].abs().max()))
    print(f"order={order}: forward {ts[0]:7.2f} ms, forward + backward {ts[1]:8.2f} ms ({ts[1] / ts[0]:.1f} x its forward){extra}", flush=True)
