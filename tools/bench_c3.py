"""BASELINE.json configs[2]: SVGP inducing-tensor path, Kzz + Kzx + Kxx-diag, T=512, N=16384, L=50, d=6, num_levels=4.
Times K_tens_n_seq_covs for RBF / linear, with / without increments, through each Kzx kernel the library has:
the tile kernel (tvs_tile_kernel.hpp) with its levels in 1, 2 or 3 sets (option tvs_tile_nw; rounds 2-4: waves per workgroup) and the older
tensor-lane kernel (tvs_tile = 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels, _lib
T, N, L, d, M = 512, 16384, 50, 6, 4
rng = np.random.default_rng(0)
X = torch.as_tensor(np.cumsum(0.2 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1), device="cuda:0")
ctx = _lib.context(0, torch.cuda.current_stream().cuda_stream)
variants = (("tile sets=auto", -1, 0), ("tile sets=1", -1, 1), ("tile sets=2", -1, 2), ("tile sets=3", -1, 3), ("tensor lanes (round 1)", 0, 0))
for base in ("rbf", "linear", "matern32"):
    for incr in (False, True):
        Z = torch.as_tensor(rng.standard_normal((M * (M + 1) // 2, T, 2, d) if incr else (M * (M + 1) // 2, T, d)), device="cuda:0")
        cls = {"rbf": kernels.SignatureRBF, "linear": kernels.SignatureLinear, "matern32": kernels.SignatureMatern32}[base]
        kern = cls(L * d, d, M, lengthscales=d ** 0.5)
        ref = None
        for name, tile, nw in variants:
            ctx.set_option("tvs_tile", tile); ctx.set_option("tvs_tile_nw", nw)
            out = kern.K_tens_n_seq_covs(Z, X, increments=incr); torch.cuda.synchronize()
            kzx = out[1]
            if ref is None: ref = kzx.clone()
            dev = float((kzx - ref).abs().max() / ref.abs().max())
            reps = 10
            ctx.timing_reset(); t0 = time.perf_counter()
            for _ in range(reps): kern.K_tens_n_seq_covs(Z, X, increments=incr)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
            ms, n, pairs = ctx.timing_get()
            bp = L * d * 8 + (2 if incr else 1) * (M * (M + 1) // 2) * d * 8 + 8
            print(f"C3 {base:8s} incr={int(incr)} {name:24s}: {dt*1e3:6.2f} ms per call (Kzz+Kzx+Kxx-diag), Kzx kernel {ms/reps:6.2f} ms; "
                  f"Kzx stream frac {T*N*bp/(ms/reps*1e-3)/8e12:.3f}; max dev vs first variant {dev:.1e}", flush=True)
ctx.set_option("tvs_tile", -1); ctx.set_option("tvs_tile_nw", 0)
