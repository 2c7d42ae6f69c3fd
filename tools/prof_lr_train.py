import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gpsig_amd import autodiff, kernels
N, T = int(sys.argv[1]), int(sys.argv[2])
L, d, M = 50, 6, 4
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
X = torch.tensor(np.cumsum(0.2 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1), device=dev)
Z = torch.tensor(rng.standard_normal((M * (M + 1) // 2, T, d)), device=dev, requires_grad=True)
kern = kernels.SignatureRBF(L * d, d, M, lengthscales=d ** 0.5, low_rank=True, num_components=50, rank_bound=50)
kern.rng = np.random.default_rng(1)
mod = autodiff.SignatureKernelModule(kern, device=dev)
def tm(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
lt = M * (M + 1) // 2
print("draw_low_rank (host)", tm(lambda: mod.draw_low_rank(lt * T + N * L)))
dr = mod.draw_low_rank(lt * T + N * L)
print("covs fwd with given draw", tm(lambda: mod.K_tens_n_seq_covs(Z, X, lr=dr)))
Xs = mod.scale_sequences(mod._seq3(X)); Zs = mod.scale_tensors(Z)
pool = torch.cat([Zs.reshape(-1, d), Xs.reshape(-1, d)], 0)
print("scope init (gather, kappa, eigh)", tm(lambda: autodiff._LowRankScope(mod, pool, dr)))
sc = autodiff._LowRankScope(mod, pool, dr)
def seq():
    sc._seq = {}; return sc.seq(Xs)
def tens():
    sc._tens = {}; return sc.tens(Zs, False)
print("seq()", tm(seq)); print("tens()", tm(tens))
def step():
    mod.zero_grad(); Z.grad = None
    Kzz, Kzx, Kxx = mod.K_tens_n_seq_covs(Z, X, lr=dr)
    (Kzz.sum() + Kzx.sum() + Kxx.sum()).backward()
print("fwd+bwd with given draw", tm(step))
