"""The device-side draw of the low-rank random objects (gpsig_lr_draw) at BASELINE configs[2]'s shape, 20 times: run under
rocprofv3 --kernel-trace --stats to see where a draw's time goes (tools/gpu_prof_draw.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels
N, L, d, M, T = 16384, 50, 6, 4, 512
rng = np.random.default_rng(0)
X = torch.as_tensor(np.cumsum(0.2 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1), device="cuda:0")
Z = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, d)), device="cuda:0")
for sp in (sys.argv[1:] or ["sqrt"]):
    kern = kernels.SignatureRBF(L * d, d, M, lengthscales=float(np.sqrt(d)), low_rank=True, num_components=50, sparsity=sp)
    kern.rng = np.random.default_rng(3)
    kern.draw_low_rank(X=X, Z=Z, _implicit=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): kern.draw_low_rank(X=X, Z=Z, _implicit=True)
    torch.cuda.synchronize()
    print(sp, "draw: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
