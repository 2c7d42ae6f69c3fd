import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpsig_amd import kernels, _lib
from oracle import sigkern_oracle as O
np.set_printoptions(linewidth=220, precision=4, suppress=False)
rng = np.random.default_rng(0)
def rel(a, b): return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)
for (n1, L1, n2, L2) in [(3, 4, 2, 6), (2, 6, 3, 4), (9, 3, 5, 40), (5, 40, 9, 3)]:
    d, M = 2, 2
    X = rng.standard_normal((n1, L1 * d)); Y = rng.standard_normal((n2, L2 * d))
    k = kernels.SignatureLinear(L1 * d, d, M, normalization=False)
    ko = O.SignatureKernelOracle(L1 * d, d, M, base="linear", normalization=False)
    got = k.K(X, Y, return_levels=True); want = ko.K(X, Y, return_levels=True)
    print(f"--- N1={n1} L1={L1} N2={n2} L2={L2}  rel {rel(got, want):.2e}")
    if rel(got, want) > 1e-9:
        print("got[1]\n", got[1], "\nwant[1]\n", want[1], "\nwant[1] of swapped args transposed\n", ko.K(Y.reshape(n2,-1), X, return_levels=True)[1].T if False else "")
