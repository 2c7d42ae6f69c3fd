"""Forward + backward of one full Gram (profiling aid): python tools/bench_grad_gram.py [N] [base] [reps] [L] [d] [M]
GPSIG_OPTIONS="sig_features_grad=0" keeps SignatureLinear's reverse pass on the pair kernels (A/B against the feature route)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import autodiff, kernels

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base = sys.argv[2] if len(sys.argv) > 2 else "rbf"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = int(sys.argv[4]) if len(sys.argv) > 4 else 64
d = int(sys.argv[5]) if len(sys.argv) > 5 else 8
M = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
if os.environ.get("GPSIG_GRAD_IMPL"):
    from gpsig_amd import _lib
    _lib.context(0, torch.cuda.current_stream(dev).cuda_stream).set_option("grad_impl", int(os.environ["GPSIG_GRAD_IMPL"]))
X = torch.tensor(rng.standard_normal((N, L * d)), device=dev)
kern = (kernels.SignatureRBF if base == "rbf" else kernels.SignatureLinear)(L * d, d, M, lengthscales=(d ** 0.5 if base == "rbf" else 1.0))
mod = autodiff.SignatureKernelModule(kern, device=dev)
mod.sum_route = os.environ.get("GPSIG_SUM_ROUTE", "1") != "0"       # 0: level primitives + torch ops for the normalisation and the level sum
W = torch.tensor(rng.standard_normal((N, N)), device=dev)
def step():
    mod.zero_grad(); (mod.K(X) * W).sum().backward()
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): step()
torch.cuda.synchronize()
print(f"full Gram N={N} L={L} d={d} M={M} {base} [{os.environ.get('GPSIG_OPTIONS', '')}{'' if mod.sum_route else ' level primitives'}]: forward+backward {(time.perf_counter() - t0) / reps * 1e3:.1f} ms")
