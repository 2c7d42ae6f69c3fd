"""GPU check of the fused reverse kernel of the RBF sequence-vs-sequence Gram (grad_fused_kernel.hpp): the planner's choice
(grad_impl 0: the fused kernel where it is built) against the stored-lattice kernels (grad_impl 1) and the sweeps with Lam through
HBM (grad_impl 4), then the bench's grad-c2shape-n1024-rbf line with either.  Run on the GPU box: python tools/gpu_fused_check.py"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gpsig_amd import _lib  # noqa: E402
from gpsig_amd.autodiff import _Spec  # noqa: E402


def vp(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def main():
    rng = np.random.default_rng(3)
    ctx = _lib.context(0, 0)
    ctx.set_pointer_mode(_lib.PTR_HOST)
    worst = 0.0
    shapes = [(5, 6, 6, 64, 64, 8, "sym"), (4, 9, 7, 20, 31, 3, "cross"), (2, 13, 13, 5, 5, 1, "sym"), (6, 5, 9, 33, 64, 7, "cross"), (3, 1, 1, 2, 2, 4, "sym"),
              (5, 70, 70, 17, 17, 8, "sym"), (4, 3, 130, 64, 9, 5, "cross"), (5, 37, 37, 64, 64, 8, "sym"), (4, 5, 4, 2, 9, 3, "cross"), (3, 40, 2, 100, 50, 6, "cross")]
    for base in ("rbf", "matern12", "matern32", "matern52"):
      for (M, N1, N2, L1, L2, d, kind) in (shapes if base == "rbf" else shapes[:4]):
        X = np.cumsum(rng.standard_normal((N1, L1, d)) * 0.3, 1)
        Y = np.cumsum(rng.standard_normal((N2, L2, d)) * 0.3, 1) if kind == "cross" else None
        G = rng.standard_normal((M + 1, N1, N2 if kind == "cross" else N1))
        keep = []
        p = _Spec(base, M, True, 0.0, order=1).params(d, 0.0, keep)
        res = []
        for impl in (1, 0, 4):
            ctx.set_option("grad_impl", impl)
            gX, gY = np.empty_like(X), (None if Y is None else np.empty_like(Y))
            ctx.call("gpsig_seq_gram_levels_grad", p, vp(X), vp(Y), N1, N2 if Y is not None else N1, L1, L2 if Y is not None else L1, vp(G), vp(gX), vp(gY), None)
            res.append((gX, gY))
        ctx.set_option("grad_impl", 0)
        e = [rel(res[1][0], res[0][0]), rel(res[2][0], res[0][0])]
        if Y is not None:
            e += [rel(res[1][1], res[0][1]), rel(res[2][1], res[0][1])]
        print(f"{base} M={M} N1={N1} N2={N2} L1={L1} L2={L2} d={d} {kind}: planner vs stored {e[0]:.2e}" + (f" (y {e[2]:.2e})" if Y is not None else "")
              + f"; Lam-through-HBM vs stored {e[1]:.2e}", flush=True)
        worst = max(worst, e[0], e[2] if Y is not None else 0.0)
    print("worst planner-vs-stored", worst, flush=True)

    import math
    import torch
    from gpsig_amd import autodiff, kernels
    dev = torch.device("cuda:0")
    N, L, D, M = 1024, 64, 8, 5
    X = torch.tensor(np.random.default_rng(0).standard_normal((N, L * D)), device=dev)
    W = torch.tensor(np.random.default_rng(1).standard_normal((N, N)), device=dev)
    dctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
    for cls in (kernels.SignatureRBF, kernels.SignatureMatern32):
      kern = cls(L * D, D, M, lengthscales=math.sqrt(D))
      mod = autodiff.SignatureKernelModule(kern, device=dev)
      grads = {}
      for impl in (0, 4, 0):
        dctx.set_option("grad_impl", impl)

        def step():
            mod.zero_grad()
            (mod.K(X) * W).sum().backward()
        for _ in range(2):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / 5 * 1e3
        g = [q.grad.detach().cpu().numpy().copy() for q in mod.parameters() if q.grad is not None]
        grads[impl] = g
        print(f"grad-c2shape-n1024 {cls.__name__} grad_impl={impl}: {ms:.2f} ms per forward + backward", flush=True)
      dctx.set_option("grad_impl", 0)
      for a, b in zip(grads[0], grads[4]):
        print("hyper-parameter gradient, planner vs Lam-through-HBM:", rel(a, b))
    return 0 if worst < 1e-9 else 1


if __name__ == "__main__":
    sys.exit(main())
