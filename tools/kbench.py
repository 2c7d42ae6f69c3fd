"""Quick kernel A/B: time the C2 Gram (N=4096, L=64, d=8, M=5) a few times with the current library and options."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels, _lib
N, L, D, M = 4096, 64, 8, 5
base = sys.argv[1] if len(sys.argv) > 1 else "linear"
X = torch.as_tensor(np.random.default_rng(0).standard_normal((N, L * D)), device="cuda:0")
kern = (kernels.SignatureLinear if base == "linear" else kernels.SignatureRBF)(L * D, D, M, lengthscales=1.0 if base == "linear" else D ** 0.5)
ctx = _lib.context(0, torch.cuda.current_stream().cuda_stream)
for opts in ([("glds", 0)], [("glds", 1)], [("glds", 1), ("max_run", 256)], [("glds", 1), ("max_run", 128)], [("glds", 1), ("max_run", 64)], [("glds", 1), ("max_run", 32)], [("glds", 1), ("max_run", 16)]):
    for k, v in (("glds", 0), ("max_run", 0)): ctx.set_option(k, v)
    for k, v in opts: ctx.set_option(k, v)
    kern.K(X); torch.cuda.synchronize()
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(5): out = kern.K(X)
    torch.cuda.synchronize()
    ms, n, pairs = ctx.timing_get()
    print(f"{os.environ.get('GPSIG_LIB','default')[-22:]:22s} {base} {opts}: kernel {ms/n:.2f} ms/launch, step {(time.perf_counter()-t0)/5*1e3:.2f} ms, frac {N*N*8200/(ms/n*1e-3)/8e12:.3f}, diag ok {abs(float(out[5,5])-6)<1e-9}")
