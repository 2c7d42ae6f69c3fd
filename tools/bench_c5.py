"""BASELINE.json configs[4]: N=2048, L=128, d=16, num_levels=6, float32 -- every float32 pair-kernel variant the library has:
the one-sequence kernels (pk2 = 0) and seq_pk2_kernel with one / two y sequences per pair group (f32_pack) and one / four
wavefronts per workgroup on one x ring (f32_waves).  Prints the kernel time and the deviation from the first variant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels, _lib
N, L, d, M = 2048, 128, 16, 6
rng = np.random.default_rng(0)
X = torch.as_tensor(np.cumsum(0.1 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1).astype(np.float32), device="cuda:0")
ctx = _lib.context(0, torch.cuda.current_stream().cuda_stream)
variants = [("one-sequence kernels (round 1)", 0, 2, 1), ("pk2: 2 y / group, 1 wave", 2, 2, 1), ("pk2: 2 y / group, 4 waves / ring", 2, 2, 4),
            ("pk2 code, 1 y / group (scalar), 1 wave", 2, 1, 1), ("pk2 code, 1 y / group (scalar), 4 waves / ring", 2, 1, 4)]
for base in ("rbf", "linear"):
    kern = (kernels.SignatureRBF if base == "rbf" else kernels.SignatureLinear)(L * d, d, M, lengthscales=d ** 0.5)
    ref = None
    for name, pk2, pack, waves in variants:
        ctx.set_option("pk2", pk2); ctx.set_option("f32_pack", pack); ctx.set_option("f32_waves", waves)
        out = kern.K(X); torch.cuda.synchronize()
        if ref is None: ref = out.double()
        dev = float((out.double() - ref).abs().max() / ref.abs().max())
        reps = 5
        ctx.timing_reset(); t0 = time.perf_counter()
        for _ in range(reps): kern.K(X)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        ms, n, pairs = ctx.timing_get()
        print(f"C5 fp32 {base:6s} {name:48s}: {dt*1e3:6.2f} ms per K(X), pair kernel {ms/reps:6.2f} ms; max dev vs first variant {dev:.1e}", flush=True)
ctx.set_option("pk2", 1); ctx.set_option("f32_pack", 2); ctx.set_option("f32_waves", 0)
