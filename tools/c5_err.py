import sys, numpy as np, torch
sys.path.insert(0, ".")
from gpsig_amd import kernels as K, _lib
from oracle import sigkern_oracle as O
rng = np.random.default_rng(0)
N, L, d, M = 2048, 128, 16, 6
X = np.cumsum(0.1 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1).astype(np.float32)
kx = K.SignatureRBF(L * d, d, M, lengthscales=np.sqrt(d) * np.ones(d))
ko = O.SignatureKernelOracle(L * d, d, M, base="rbf", lengthscales=np.sqrt(d) * np.ones(d))
Xd = torch.as_tensor(X, device="cuda:0")
ctx = _lib.context(0, 0)
blocks = ((np.arange(0, 12), np.arange(0, 12)), (np.arange(1000, 1008), np.arange(2040, 2048)), (np.array([3, 1025, 2047]), np.array([0, 1023, 1024, 1026])))
wants = [ko.K(X[r].astype(np.float64), X[c].astype(np.float64)) for r, c in blocks]
for pk2 in (1, 0):
    ctx.set_option("pk2", pk2)
    G = kx.K(Xd).cpu().numpy().astype(np.float64)
    G64 = kx.K(Xd.double()[:64]).cpu().numpy()
    for (r, c), w in zip(blocks, wants):
        e = np.abs(G[np.ix_(r, c)] - w)
        print("pk2", pk2, "block", r[:2], c[:2], "max abs err", e.max(), "rel to max", e.max() / np.abs(w).max(), "argmax", np.unravel_index(e.argmax(), e.shape), "want there", w.flat[e.argmax()])
