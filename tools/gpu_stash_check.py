"""GPU check of the stash route of the RBF sequence Gram's gradient (gpsig_seq_gram_levels_stash + _grad_stash through
autodiff._SeqGramLevels): gradients with the stash against the recompute route (option grad_stash_mb = 0), then the timing of both."""
import math
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from gpsig_amd import _lib, autodiff  # noqa: E402
from gpsig_amd.autodiff import _SeqGramLevels, _Spec  # noqa: E402


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def main():
    dev = torch.device("cuda:0")
    ctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(11)
    worst = 0.0
    for (M, N1, N2, L1, L2, d, kind) in [(5, 64, 64, 64, 64, 8, "sym"), (4, 9, 7, 20, 31, 3, "cross"), (5, 13, 13, 5, 5, 2, "sym"), (4, 130, 130, 33, 33, 5, "sym"),
                                           (5, 37, 41, 64, 64, 8, "cross"), (5, 300, 300, 17, 17, 8, "sym"), (3, 10, 10, 12, 12, 4, "sym"), (5, 301, 301, 9, 9, 4, "sym")]:
        X = torch.tensor(np.cumsum(rng.standard_normal((N1, L1, d)) * 0.3, 1), device=dev)
        Y = torch.tensor(np.cumsum(rng.standard_normal((N2, L2, d)) * 0.3, 1), device=dev) if kind == "cross" else None
        G = torch.tensor(rng.standard_normal((M + 1, N1, N2 if kind == "cross" else N1)), device=dev)
        spec = _Spec("rbf", M, True, 0.0, order=1)
        res = []
        took = []
        for mb in (4096, 0):
            ctx.set_option("grad_stash_mb", mb)
            Xg = X.clone().requires_grad_(True)
            Yg = None if Y is None else Y.clone().requires_grad_(True)
            lev = _SeqGramLevels.apply(Xg, Yg, None, spec)
            took.append(lev.grad_fn.stash is not None if hasattr(lev.grad_fn, "stash") else None)
            (lev * G).sum().backward()
            res.append((lev.detach().clone(), Xg.grad.clone(), None if Yg is None else Yg.grad.clone()))
        ctx.set_option("grad_stash_mb", 4096)
        e = [rel(res[0][0], res[1][0]), rel(res[0][1], res[1][1])] + ([rel(res[0][2], res[1][2])] if Y is not None else [])
        print(f"M={M} N1={N1} N2={N2} L1={L1} L2={L2} d={d} {kind}: stash kept {took[0]} / {took[1]}; levels {e[0]:.2e}, dX {e[1]:.2e}" + (f", dY {e[2]:.2e}" if Y is not None else ""), flush=True)
        worst = max(worst, *e)
    print("worst", worst, flush=True)
    from gpsig_amd import kernels
    N, L, D, M = 1024, 64, 8, 5
    X = torch.tensor(np.random.default_rng(0).standard_normal((N, L * D)), device=dev)
    W = torch.tensor(np.random.default_rng(1).standard_normal((N, N)), device=dev)
    mod = autodiff.SignatureKernelModule(kernels.SignatureRBF(L * D, D, M, lengthscales=math.sqrt(D)), device=dev)
    grads = {}
    for mb in (4096, 0, 4096):
        ctx.set_option("grad_stash_mb", mb)

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
        grads[mb] = [q.grad.detach().clone() for q in mod.parameters() if q.grad is not None]
        print(f"grad-c2shape-n1024-rbf grad_stash_mb={mb}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per forward + backward", flush=True)
    ctx.set_option("grad_stash_mb", 4096)
    for a, b in zip(grads[4096], grads[0]):
        print("hyper-parameter gradient, stash vs recompute:", rel(a, b))
    return 0 if worst < 1e-9 else 1


if __name__ == "__main__":
    sys.exit(main())
