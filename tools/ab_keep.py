"""A/B on one box: pair kernels with the accumulators cleared through SeqLane::keep (keep_reset = 1) against the explicit reset in
the pair-boundary block (0), alternating, at BASELINE configs[1] (linear / RBF, fp64) and configs[4] (RBF, fp32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels, _lib
rng = np.random.default_rng(0)
ctx = _lib.context(0, torch.cuda.current_stream().cuda_stream)
cases = []
X2 = torch.as_tensor(rng.standard_normal((4096, 64 * 8)), device="cuda:0")
cases.append(("c2 linear fp64", kernels.SignatureLinear(64 * 8, 8, 5), X2))
cases.append(("c2 rbf    fp64", kernels.SignatureRBF(64 * 8, 8, 5, lengthscales=8 ** 0.5), X2))
X5 = torch.as_tensor(np.cumsum(0.1 * rng.standard_normal((2048, 128, 16)), axis=1).reshape(2048, -1).astype(np.float32), device="cuda:0")
cases.append(("c5 rbf    fp32", kernels.SignatureRBF(128 * 16, 16, 6, lengthscales=4.0), X5))
cases.append(("c5 linear fp32", kernels.SignatureLinear(128 * 16, 16, 6), X5))
for name, kern, X in cases:
    outs = {}
    for rnd in range(3):
        for keep in (0, 1):
            ctx.set_option("keep_reset", keep)
            out = kern.K(X); torch.cuda.synchronize()
            outs[keep] = out
            ctx.timing_reset()
            for _ in range(5): kern.K(X)
            torch.cuda.synchronize()
            ms, n, _ = ctx.timing_get()
            print(f"{name} round {rnd} keep_reset={keep}: pair kernel {ms / 5:7.3f} ms", flush=True)
    print(f"{name}: bitwise equal outputs: {bool(torch.equal(outs[0], outs[1]))}")
ctx.set_option("keep_reset", 1)
