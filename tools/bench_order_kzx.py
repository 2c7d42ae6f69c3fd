"""Kzx at BASELINE configs[2]'s size (T = 512 inducing tensors, N = 16,384, L = 50, d = 6, num_levels = 4) for the first- and
higher-order algorithms (signature_algs.py:101-160), SignatureLinear and SignatureRBF."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import kernels  # noqa: E402

T, N, L, d, M = 512, 16384, 50, 6, 4
rng = np.random.default_rng(0)
X = torch.as_tensor(rng.standard_normal((N, L * d)) * 0.3, device="cuda:0")
Z = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, d)) * 0.3, device="cuda:0")
from gpsig_amd import _lib  # noqa: E402
from oracle import sigkern_oracle as O  # noqa: E402
ctx = _lib.context(0, torch.cuda.current_stream(torch.device("cuda:0")).cuda_stream)
Zi = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, 2, d)) * 0.3, device="cuda:0")
for cls, base in ((kernels.SignatureLinear, "linear"), (kernels.SignatureRBF, "rbf")):
    for inc in (False, True):
        for order in (1, 2, 4):
            k = cls(L * d, d, M, order=order)
            Zu = Zi if inc else Z
            res = {}
            for tile in ((-1, 0) if (base == "rbf" and order > 1) else (-1,)):      # round 6: higher-order chains in the tile kernel; 0 = the older mappings
                ctx.set_option("tvs_tile", tile)
                G = k.K_tens_vs_seq(Zu, X, increments=inc); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    G = k.K_tens_vs_seq(Zu, X, increments=inc)
                torch.cuda.synchronize()
                res[tile] = ((time.perf_counter() - t0) / 3 * 1e3, G)
            ctx.set_option("tvs_tile", -1)
            ko = O.SignatureKernelOracle(L * d, d, M, base=base, order=order)
            want = ko.K_tens_vs_seq(Zu[:, :16].cpu().numpy(), X[:32].cpu().numpy(), increments=inc)
            err = float(np.abs(res[-1][1][:16, :32].cpu().numpy() - want).max() / np.abs(want).max())
            print(cls.__name__, "increments" if inc else "plain", "order", order, "%.2f ms" % res[-1][0],
                  ("(older mappings %.2f ms)" % res[0][0]) if 0 in res else "", "vs oracle %.1e" % err, flush=True)

# the reverse pass (round 6: the wide route's chains of every order; the forward pass that feeds it is the tile kernel's where one is built)
from gpsig_amd import autodiff  # noqa: E402
print("\nforward + backward of Kzx (autodiff module; gradient w.r.t. the inducing tensors, the lengthscales and the variances):")
for inc in (False, True):
    base_t = None
    for order in (1, 2, 4):
        mod = autodiff.SignatureKernelModule(kernels.SignatureRBF(L * d, d, M, order=order), device="cuda:0")
        Zp = (Zi if inc else Z).clone().requires_grad_(True)

        def fwd():
            with torch.no_grad():
                return mod.K_tens_vs_seq(Zp, X, increments=inc)

        def fb():
            Zp.grad = None
            mod.zero_grad(set_to_none=True)
            o = mod.K_tens_vs_seq(Zp, X, increments=inc)
            (o * o).sum().backward()
        ts = []
        for fn in (fwd, fb):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 3 * 1e3)
        base_t = base_t or ts
        print("SignatureRBF", "increments" if inc else "plain", "order", order, "forward %.2f ms, forward + backward %.2f ms (%.1f x its forward; %.2f x order 1's)"
              % (ts[0], ts[1], ts[1] / ts[0], ts[1] / base_t[1]), flush=True)
