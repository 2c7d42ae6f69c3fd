"""The evaluation path (gpsig_amd.kernels: the reference's K / Kdiag / K_tens_n_seq_covs call surface, CUDA tensors in) for every base-kernel family, float64 and float32,
orders 1 and 2: a search for slow corners.  python tools/probe_eval_families.py [N]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import kernels
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T, L, d, M = 512, 50, 6, 4
rng = np.random.default_rng(0)
X64 = torch.as_tensor(np.cumsum(rng.standard_normal((N, L, d)) * 0.2, axis=1).reshape(N, -1), device="cuda:0")
Z64 = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, 2, d)) * 0.4, device="cuda:0")
FAM = [("linear", kernels.SignatureLinear), ("cosine", kernels.SignatureCosine), ("poly", kernels.SignaturePoly), ("rbf", kernels.SignatureRBF), ("mix", kernels.SignatureMix),
       ("matern12", kernels.SignatureMatern12), ("matern32", kernels.SignatureMatern32), ("matern52", kernels.SignatureMatern52)]
def timed(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for name, cls in FAM:
    for order in (1, 2):
        for dt in (torch.float64, torch.float32):
            try:
                k = cls(L * d, d, M, order=order, lengthscales=np.sqrt(d))
                X, Z = X64.to(dt), Z64.to(dt)
                tk = timed(lambda: k.K(X))
                tc = timed(lambda: k.K_tens_n_seq_covs(Z, X, increments=True))
                td = timed(lambda: k.Kdiag(X))
                print("%-9s order %d %-7s | K(X) N=%d %8.2f ms | Kzz, Kzx, Kxx-diag %8.2f ms | Kdiag %6.2f ms" % (name, order, str(dt).split(".")[1], N, tk, tc, td), flush=True)
            except Exception as e:
                print("%-9s order %d %-7s | FAILED %s: %s" % (name, order, str(dt).split(".")[1], type(e).__name__, str(e)[:140]), flush=True)
