import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpsig_amd import kernels, _lib
from oracle import sigkern_oracle as O
np.set_printoptions(linewidth=200, precision=4, suppress=False)
rng = np.random.default_rng(0)
ctx = _lib.context(0, 0)

def rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)

def probe(N, L, d, M, base="linear", exact=1, glds=0, show=False, N2=None, L2=None):
    X = rng.standard_normal((N, L, d))
    Y = rng.standard_normal((N2, L2 or L, d)) if N2 else None
    cls = kernels.SignatureLinear if base == "linear" else kernels.SignatureRBF
    k = cls(L * d, d, M, normalization=False, lengthscales=None)
    ko = O.SignatureKernelOracle(L * d, d, M, base=base, normalization=False, lengthscales=None)
    ctx.set_option("exact", exact); ctx.set_option("glds", glds)
    got = k._K_seq(X, Y)
    want = ko._K_seq(X, Y)
    errs = [rel(got[m], want[m]) for m in range(M + 1)]
    print(f"N={N} L={L} d={d} M={M} {base} exact={exact} glds={glds} N2={N2}: per-level rel err", " ".join(f"{e:.1e}" for e in errs), "nan:", int(np.isnan(got).sum()))
    if show:
        bad = np.abs(got[1] - want[1]) > 1e-9 * np.abs(want[1]).max()
        print("level-1 bad mask:\n", bad.astype(int))
        print("got[1][:4,:4]\n", got[1][:4, :4], "\nwant\n", want[1][:4, :4])
        print("got[M][:4,:4]\n", got[M][:4, :4], "\nwant\n", want[M][:4, :4])
    ctx.set_option("exact", 1); ctx.set_option("glds", 0)

probe(4, 5, 3, 2, show=True)
probe(4, 5, 3, 2, N2=4, show=True)
probe(9, 13, 3, 4, exact=0)
probe(9, 13, 3, 4, exact=1)
probe(9, 32, 3, 4, exact=0)
probe(9, 32, 3, 4, exact=1)
probe(9, 32, 3, 4, exact=1, glds=1)
probe(9, 64, 8, 5, exact=1)
probe(9, 64, 8, 5, exact=0)
probe(9, 13, 3, 4, base="rbf", exact=0)
probe(9, 13, 3, 4, base="rbf", exact=0, N2=5)
