#!/usr/bin/env python3
"""Low-rank mode (gpsig/low_rank_calculations.py, signature_algs.py:162-222, kernels.py:236-311) at the benchmark shapes.

    python tools/bench_lr.py [--config c3|c2] [--base rbf|linear] [--components 50] [--rank 50] [--sparsity sqrt] [--fused 1|0]

c3: the SVGP inducing-tensor path of BASELINE configs[2] in low-rank mode -- K_tens_n_seq_covs(Z, X), T=512, N=16384, L=50, d=6, M=4.
c2: K(X) at BASELINE configs[1]'s shape, N=4096, L=64, d=8, M=5.
Inputs resident in HBM, random objects (landmarks, whitening, sketches) drawn once outside the timed region and handed in, as
the parity tests do; `with_draw` is the same evaluation with a fresh draw per call (landmark gather, rocSOLVER eigendecomposition,
host-side sketches: what the reference does per TF session run).  Per-stage times are wall-clock around synchronised C-ABI calls.
Prints one JSON line; --verify compares a sub-sample with the oracle's restatement given the same random objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SHAPES = {"c3": dict(N=16384, L=50, d=6, M=4, T=512), "c2": dict(N=4096, L=64, d=8, M=5, T=0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=sorted(SHAPES))
    ap.add_argument("--base", default="rbf", choices=["rbf", "linear"])
    ap.add_argument("--components", type=int, default=50)
    ap.add_argument("--rank", type=int, default=None)
    ap.add_argument("--sparsity", default="sqrt", choices=["sqrt", "log", "lin"])
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--pad", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--verify", action="store_true")
    args = ap.parse_args()
    import torch
    from gpsig_amd import _lib, kernels
    w = SHAPES[args.config]
    N, L, d, M, T = (w[k] for k in ("N", "L", "d", "M", "T"))
    rng = np.random.default_rng(0)
    Xh = np.cumsum(0.2 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1)
    Zh = np.random.default_rng(1).standard_normal((M * (M + 1) // 2, T, d)) if T else None
    dev = torch.device("cuda", 0)
    X = torch.as_tensor(Xh, device=dev)
    Z = torch.as_tensor(Zh, device=dev) if T else None
    cls = kernels.SignatureLinear if args.base == "linear" else kernels.SignatureRBF
    ls = 1.0 if args.base == "linear" else float(np.sqrt(d))
    kern = cls(L * d, d, M, lengthscales=ls, low_rank=True, num_components=args.components, rank_bound=args.rank, sparsity=args.sparsity)
    kern.rng = np.random.default_rng(3)
    ctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_option("lr_fused", args.fused)
    ctx.set_option("lr_fused_variant", args.variant)
    ctx.set_option("lr_fused_pad", args.pad)
    st = kern.draw_low_rank(X=X, Z=Z)                 # CUDA tensors: drawn on the device (gpsig_lr_draw)
    sth = st.export() if hasattr(st, "export") else st   # host copies: entry counts, the oracle's input

    def timed(fn, steps=args.steps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, out

    if T:
        # K_tens_n_seq_covs with the draw handed in: the same calls kernels.py makes, the random objects fixed
        def evaluate():
            L_ = kernels._launch_f64(Z, X)
            p = kern._params(L_.keep)
            lr = st.as_c(L_.keep)
            PZ, pz, t = kern._lr_features(L_, p, lr, Z, tensors=True)
            PX, px, n = kern._lr_features(L_, p, lr, X)
            Kzz, ozz = L_.out((t, t))
            L_.ctx.call("gpsig_lr_kernel", p, lr, pz, None, t, t, 0, 0, 0, ozz)
            Kzx, ozx = L_.out((t, n))
            L_.ctx.call("gpsig_lr_kernel", p, lr, pz, px, t, n, 0, 1, 0, ozx)
            return Kzz, Kzx, kern.Kdiag(X, presliced=True, lr_state=st)
        fresh = lambda: kern.K_tens_n_seq_covs(Z, X)                      # noqa: E731
        pairs = float(T) * N
    else:
        evaluate = lambda: kern.K(X, lr_state=st)                         # noqa: E731
        fresh = lambda: kern.K(X)                                         # noqa: E731
        pairs = float(N) * N
    ms, out = timed(evaluate)
    ms_fresh, _ = timed(fresh, steps=3)
    # stages
    L_ = kernels._launch_f64(X)
    p = kern._params(L_.keep)
    lr = st.as_c(L_.keep)
    ms_seq, (PX, px, n) = timed(lambda: kern._lr_features(L_, p, lr, X))
    stages = {"seq_features_ms": ms_seq}
    if T:
        Lz = kernels._launch_f64(Z, X)
        ms_tens, (PZ, pz, t) = timed(lambda: kern._lr_features(Lz, p, lr, Z, tensors=True))
        Kzx, ozx = Lz.out((t, n))
        ms_gemm, _ = timed(lambda: Lz.ctx.call("gpsig_lr_kernel", p, lr, pz, px, t, n, 0, 1, 0, ozx))
        stages.update(tens_features_ms=ms_tens, kzx_product_ms=ms_gemm)
    else:
        Kxx, oxx = L_.out((n, n))
        ms_gemm, _ = timed(lambda: L_.ctx.call("gpsig_lr_kernel", p, lr, px, None, n, n, 1, 1, 0, oxx))
        stages.update(gram_product_ms=ms_gemm)
    F = 1 + args.components + (M - 1) * (args.rank or args.components)
    nnz = [int(s.val.shape[0]) for s in sth.sketches]
    ms_draw, _ = timed(lambda: kern.draw_low_rank(X=X, Z=Z, _implicit=True))
    kern.device_draw = False
    ms_draw_host, _ = timed(lambda: kern.draw_low_rank(X=X, Z=Z), steps=3)
    kern.device_draw = True
    l = L - 1
    res = {"what": f"low-rank mode, {args.config} shape: " + (f"K_tens_n_seq_covs, T={T} inducing tensors, " if T else "K(X), ") +
                   f"N={N}, L={L}, d={d}, num_levels={M}, Signature{'Linear' if args.base == 'linear' else 'RBF'}, fp64, "
                   f"num_components={args.components}, rank_bound={args.rank or args.components}, sparsity={args.sparsity}",
           "ms_per_evaluation": ms, "entries_per_s": pairs / (ms * 1e-3), "ms_with_fresh_draw": ms_fresh, "ms_draw_on_device": ms_draw, "ms_draw_on_host_round2": ms_draw_host, "stages": stages,
           "fused_feature_kernel": bool(args.fused), "feature_width": F, "sketch_nnz_per_level": nnz,
           # the fused kernel's model: per sequence and sketch entry two 512-byte LDS reads per 64 time steps (lr_fused_kernel.hpp)
           "seq_features_lds_bytes": float(N) * ((l + 63) // 64) * 64 * 8 * (2 * sum(nnz) + args.components ** 2),
           "seq_features_flops": float(N) * l * (3 * sum(nnz) + 2 * args.components ** 2 + args.components * (2 * d + 20))}
    res["seq_features_lds_GBps"] = res["seq_features_lds_bytes"] / (ms_seq * 1e-3) / 1e9
    if args.verify:
        from oracle import sigkern_oracle as O
        ko = O.SignatureKernelOracle(L * d, d, M, base=args.base, lengthscales=ls)
        lo = O.LowRankOracle(ko, sth.landmarks, sth.jitter_diag, sth.sketches)
        ns, ts = 20, 12
        if T:
            want = lo.K_tens_vs_seq(Zh[:, :ts], Xh[:ns])
            got = out[1][:ts, :ns].cpu().numpy()
        else:
            want = lo.K(Xh[:ns])
            got = out[:ns, :ns].cpu().numpy()
        res["rel_err_vs_oracle_same_randomness"] = float(np.abs(got - want).max() / np.abs(want).max())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
