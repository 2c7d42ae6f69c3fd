import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, ctypes as C
from oracle import sigkern_oracle_torch as OT
from gpsig_amd import _lib
from gpsig_amd.autodiff import _Spec
ctx = _lib.context(0, 0); ctx.set_pointer_mode(_lib.PTR_HOST)
_P = C.POINTER(C.c_double)
for (M, d, order) in ((2, 32, 2), (2, 32, 1), (2, 16, 2)):
    rng = np.random.default_rng(1)
    N, L = 5, 6
    X = rng.standard_normal((N, L, d)) * 0.6
    G = rng.standard_normal((M + 1, N, N))
    kt = OT.SignatureKernelTorchOracle(d, M, "linear", p0=None, p1=0.0, difference=True, order=order)
    tX = torch.tensor(X, requires_grad=True)
    lev = kt.K_seq_levels(tX, None)
    (lev * torch.tensor(G)).sum().backward()
    keep = []
    p = _Spec("linear", M, True, 0.0, order=order).params(d, 0.0, keep)
    # forward levels through the library
    out = np.empty((M + 1, N, N))
    ctx.set_option("sig_features", 1)
    ctx.call("gpsig_seq_gram_levels", p, C.c_void_p(X.ctypes.data), None, N, N, L, L, C.c_void_p(out.ctypes.data))
    ctx.set_option("sig_features", -1)
    print(M, d, order, "forward levels rel", float(np.abs(out - lev.detach().numpy()).max() / np.abs(lev.detach().numpy()).max()))
    ctx.set_option("sig_features_grad", 1)
    gX, gb = np.full_like(X, np.nan), np.zeros(2)
    ctx.call("gpsig_seq_gram_levels_grad", p, C.c_void_p(X.ctypes.data), None, N, N, L, L, C.c_void_p(G.ctypes.data), C.c_void_p(gX.ctypes.data), None, gb.ctypes.data_as(_P))
    ctx.set_option("sig_features_grad", -1)
    ref = tX.grad.numpy()
    err = np.abs(gX - ref).max(axis=(0, 1)) / np.abs(ref).max()
    print("   per-component err", np.array2string(err, precision=1, max_line_width=250))
    # cross
    for (N1, N2, L1, L2) in ((6, 5, 8, 5), (3, 4, 2, 9)):
        X = rng.standard_normal((N1, L1, d)) * 0.6; Y = rng.standard_normal((N2, L2, d)) * 0.6
        G = rng.standard_normal((M + 1, N1, N2))
        tX = torch.tensor(X, requires_grad=True); tY = torch.tensor(Y, requires_grad=True)
        (kt.K_seq_levels(tX, tY) * torch.tensor(G)).sum().backward()
        for route in (1, 0):
            ctx.set_option("sig_features_grad", route)
            gX, gY = np.full_like(X, np.nan), np.full_like(Y, np.nan)
            ctx.call("gpsig_seq_gram_levels_grad", p, C.c_void_p(X.ctypes.data), C.c_void_p(Y.ctypes.data), N1, N2, L1, L2, C.c_void_p(G.ctypes.data),
                     C.c_void_p(gX.ctypes.data), C.c_void_p(gY.ctypes.data), gb.ctypes.data_as(_P))
            ctx.set_option("sig_features_grad", -1)
            ex = np.abs(gX - tX.grad.numpy()).max(axis=(0, 1)) / np.abs(tX.grad.numpy()).max()
            ey = np.abs(gY - tY.grad.numpy()).max(axis=(0, 1)) / np.abs(tY.grad.numpy()).max()
            print("   cross route", route, N1, N2, L1, L2, "err by component: x", np.array2string(ex, precision=1, max_line_width=400), " y", np.array2string(ey, precision=1, max_line_width=400))
