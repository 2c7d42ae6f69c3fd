"""Host-pointer (numpy in, numpy out) evaluation of BASELINE configs[1] against the device-resident one: where the extra time goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels, _lib
N, L, d, M = 4096, 64, 8, 5
X = np.random.default_rng(0).standard_normal((N, L * d))
Xd = torch.as_tensor(X, device="cuda:0")
kern = kernels.SignatureLinear(L * d, d, M)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("device pointers            %.2f ms" % t(lambda: kern.K(Xd)))
hctx = _lib.context(0, 0)
for ps in (0, 1):
    hctx.set_option("pinned_staging", ps)
    print("host pointers, pinned=%d    %.2f ms" % (ps, t(lambda: kern.K(X))))
Kd = kern.K(Xd)
print("torch .cpu() of the result %.2f ms" % t(lambda: Kd.cpu()))
pin = torch.empty((N, N), dtype=torch.float64).pin_memory()
print("copy into pinned tensor    %.2f ms" % t(lambda: (pin.copy_(Kd, non_blocking=True), torch.cuda.synchronize())))
out = np.empty((N, N))
print("np.copyto from pinned (reused destination)  %.2f ms" % t(lambda: np.copyto(out, pin.numpy())))
print("np.empty + copy from pinned (fresh pages)   %.2f ms" % t(lambda: np.copyto(np.empty((N, N)), pin.numpy())))
