"""configs[1]'s Gram with a Matern family (profiling aid): python tools/bench_matern_fwd.py [N] [matern12|matern32|matern52] [reps]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = sys.argv[2] if len(sys.argv) > 2 else "matern32"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L, D, M = 64, 8, 5
dev = torch.device("cuda:0")
X = torch.tensor(np.cumsum(np.random.default_rng(0).standard_normal((N, L, D)) * 0.3, 1).reshape(N, -1), device=dev)
cls = {"matern12": kernels.SignatureMatern12, "matern32": kernels.SignatureMatern32, "matern52": kernels.SignatureMatern52}[base]
kern = cls(L * D, D, M, lengthscales=math.sqrt(D))
kern.K(X); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): kern.K(X)
torch.cuda.synchronize()
print(f"K(X) N={N} L={L} d={D} M={M} {base} [{os.environ.get('GPSIG_OPTIONS', '')}]: {(time.perf_counter() - t0) / reps * 1e3:.1f} ms")
