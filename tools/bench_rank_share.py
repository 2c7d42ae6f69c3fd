"""What ONE rank of an 8-rank ShardedGram at BASELINE configs[3] (N = 32,768, L = 64, d = 8, num_levels = 5) computes, timed on the one
GPU of this box: the `chunks` row-block calls of rank `--rank` (gpsig_kernel_K_symm_rows_compact), with and without keeping the feature
matrix between them.  No collective runs here; it is the compute share the scaling curve is made of."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpsig_amd import _lib, kernels, parallel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32768)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--chunks", type=int, default=4)
ap.add_argument("--base", default="linear")
a = ap.parse_args()
L, d, M = 64, 8, 5
dev = torch.device("cuda:0")
X = torch.as_tensor(np.random.default_rng(0).standard_normal((a.n, L * d)), device=dev)
kern = (kernels.SignatureLinear if a.base == "linear" else kernels.SignatureRBF)(L * d, d, M)
g = parallel.ShardedGram(kern, a.n, dev, a.rank, a.world, chunks=a.chunks)
ctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
ctx.set_pointer_mode(_lib.PTR_DEVICE)
keep = []
p = kern._params(keep)
b0 = g.bounds[a.rank]


def share(keep_features):
    ctx.set_option("sig_features_keep", 1 if keep_features else 0)
    try:
        for k in range(a.chunks):
            r0 = b0 + k * g.chunk_rows
            ctx.call("gpsig_kernel_K_symm_rows_compact", p, C.c_void_p(X.data_ptr()), a.n, L, r0, r0 + g.chunk_rows,
                     C.c_void_p(g.rows[k * g.chunk_rows:].data_ptr()))
    finally:
        ctx.set_option("sig_features_keep", 0)


for kf in (True, False):
    share(kf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        share(kf)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    pairs = float(a.n) * a.n / a.world
    print(f"rank {a.rank} of {a.world}, N={a.n}, {a.base}, {a.chunks} chunks, features kept={kf}: {ms:.1f} ms per Gram share "
          f"= {pairs / ms * 1e3:.3e} delivered pairs/s per GPU; x{a.world} = {pairs * a.world / ms * 1e3:.3e}")
