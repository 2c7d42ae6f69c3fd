"""One Kzx forward + backward at a reference shape, repeated; run under rocprofv3 --kernel-trace and read with tools/rocprof_timeline.py to see where the
time between kernels goes.   python tools/timeline_kzx.py [dataset] [fwd|fb]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import reference_shapes as RS  # noqa: E402

ds = sys.argv[1] if len(sys.argv) > 1 else "NetFlow"
mode = sys.argv[2] if len(sys.argv) > 2 else "fb"
s, model, X, Y = RS.build(ds, "cuda:0")
k = model.kernel
with torch.no_grad():
    Xs0 = k.scale_sequences(k._seq3(X, False))
    Zs0 = k.scale_tensors(model.Z)
    fac0 = torch.ones((k.kern.num_levels + 1, Xs0.shape[0]), dtype=Xs0.dtype, device=Xs0.device)
Zr, Xr = Zs0.clone().requires_grad_(mode == "fb"), Xs0.clone().requires_grad_(mode == "fb")


def it():
    if mode == "fb":
        Zr.grad = Xr.grad = None
        o = k._tvs_weighted(Zr, Xr, fac0, True)
        o.sum().backward()
    else:
        with torch.no_grad():
            k._tvs_weighted(Zr, Xr, fac0, True)


for _ in range(3):
    it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    it()
torch.cuda.synchronize()
print("%s %s: %.3f ms per iteration (10 iterations, one synchronisation at the end)" % (ds, mode, (time.perf_counter() - t0) / 10 * 1e3))
