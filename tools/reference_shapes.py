"""The reference's OWN run settings as shapes (round 6).

/root/reference/benchmarks/run_gpsig_benchmarks.py:32 trains every data set of benchmarks/datasets.json with
    num_levels=4, num_inducing=500, max_len=500, num_lags=1, increments=True
through benchmarks/models/train_gpsig.py:20-66: SignatureRBF, add_time=True (benchmarks/utils/datasets.py:34-36 -> num_features = n_features + 1),
minibatch_size=50, whiten=True, Bernoulli / MultiClass likelihood.  With num_lags=1 the state space has 2 * (n_features + 1) columns
(gpsig/kernels.py:350).  The data sets themselves are not shipped (benchmarks/datasets/download_data.sh); the numbers below (n_train, l_max,
n_features, n_classes) are the entries of datasets.json, the data synthetic paths of that shape.

For each data set this tool times, on one MI355X:
    covs fwd        Kzz, Kzx, Kxx-diag                   (autodiff.SignatureKernelModule.K_tens_n_seq_covs, no_grad)
    covs fwd+bwd    the same + the reverse pass w.r.t. Z, lengthscales, lags, gamma, variances
    step            -ELBO forward + backward of models.SVGPModule (covariances, conditional, KL, likelihood)
per route: "exact" (the exact-shape kernels of the level primitives behind the C ABI), "matrix" (base-kernel tensors by torch GEMMs in HBM, recursions
in the library), "wide" (round 6, behind the C ABI: dgemm + fused kappa / difference / recursion kernels), "auto" (what a user gets).

    python tools/reference_shapes.py [all|<dataset>[,<dataset>..]] [--routes auto,exact,matrix,wide] [--reps 5] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

# benchmarks/datasets.json (n_train, n_classes, l_max, n_features); kept here so that the tool runs where /root/reference does not exist
DATASETS = {
    "ArabicDigits": (6600, 10, 93, 13), "AUSLAN": (1140, 95, 136, 22), "CharacterTrajectories": (300, 20, 205, 3), "CMUsubject16": (29, 2, 580, 62),
    "DigitShapes": (24, 4, 98, 2), "ECG": (100, 2, 152, 2), "JapaneseVowels": (270, 9, 29, 12), "KickvsPunch": (16, 2, 841, 62),
    "LIBRAS": (180, 15, 45, 2), "NetFlow": (803, 2, 997, 4), "PEMS": (267, 7, 144, 963), "PenDigits": (300, 10, 8, 2), "Shapes": (18, 3, 98, 2),
    "UWave": (896, 8, 315, 3), "Wafer": (298, 2, 198, 6), "WalkvsRun": (28, 2, 1918, 62),
}
NUM_LEVELS, NUM_INDUCING, MAX_LEN, NUM_LAGS, MINIBATCH, VAL_SPLIT = 4, 500, 500, 1, 50, 0.2


def shape_of(name):
    n_train, n_classes, l_max, n_features = DATASETS[name]
    n_fit = n_train - int(round(VAL_SPLIT * n_train))                # train_gpsig.py:27 (val_split=0.2)
    d = n_features + 1                                               # add_time=True
    return dict(name=name, d=d, d_eff=d * (NUM_LAGS + 1), L=min(l_max, MAX_LEN), N=min(MINIBATCH, n_fit), n_fit=n_fit, classes=n_classes,
                T=NUM_INDUCING, M=NUM_LEVELS)


def build(name, device, seed=0):
    import torch
    from gpsig_amd import inducing_variables as iv, kernels, likelihoods, models
    s = shape_of(name)
    rng = np.random.default_rng(seed)
    N, L, d, T, M = s["N"], s["L"], s["d"], s["T"], s["M"]
    # normalised data with a time coordinate in the first column (datasets.py:34-36): smooth random paths of unit scale
    X = np.cumsum(rng.standard_normal((N, L, d)) / np.sqrt(L), axis=1)
    X[:, :, 0] = np.linspace(0.0, 1.0, L)[None, :]
    lt = M * (M + 1) // 2
    # utils.py:25-55: inducing tensors are pairs of consecutive observations of random sequences, tiled over the lag copies, + 0.4 * noise
    idx_n, idx_t = rng.integers(0, N, size=(lt, T)), rng.integers(0, L - 1, size=(lt, T))
    Z = np.stack([X[idx_n, idx_t], X[idx_n, idx_t + 1]], axis=2)                                  # (lt, T, 2, d)
    Z = np.tile(Z[:, :, :, None, :], (1, 1, 1, NUM_LAGS + 1, 1)).reshape(lt, T, 2, -1)
    Z = Z + 0.4 * rng.standard_normal(Z.shape)
    ls = np.sqrt(d) * np.ones(d) * 0.7                                                            # ~ utils.py:88-98 on unit-scale data
    kern = kernels.SignatureRBF(L * d, d, M, lengthscales=ls, num_lags=NUM_LAGS, order=int(os.environ.get("RS_ORDER", "1")))      # RS_ORDER: beyond the reference's runs
    feat = iv.InducingTensors(Z, M, increments=True)
    if s["classes"] == 2:
        lik, latent, Y = likelihoods.Bernoulli(), 1, rng.integers(0, 2, size=(N, 1)).astype(np.float64)
    else:
        lik, latent, Y = likelihoods.MultiClass(s["classes"]), s["classes"], rng.integers(0, s["classes"], size=(N, 1)).astype(np.float64)
    model = models.SVGPModule(kern, feat, lik, num_latent=latent, num_data=s["n_fit"], device=device)
    Xd = torch.as_tensor(X.reshape(N, -1), device=device)
    Yd = torch.as_tensor(Y, device=device)
    return s, model, Xd, Yd


def set_route(model, route):
    """auto: what a user gets.  exact: the exact-shape kernels only (beyond 64 columns the matrix route: they are not built there).
    matrix: base-kernel tensors by torch GEMMs, recursions in the library.  wide: the library's wide route wherever built."""
    from gpsig_amd import autodiff
    k = model.kernel
    if route == "auto":
        k.matrix_route = k._auto_matrix_route
        autodiff.set_wide_route(None)
    elif route == "exact":
        k.matrix_route = False
        autodiff.set_wide_route(False)
    elif route == "matrix":
        k.matrix_route = True
        autodiff.set_wide_route(False)
    elif route == "wide":
        k.matrix_route = False
        autodiff.set_wide_route(True)
    else:
        raise ValueError(route)


def timed(fn, reps, budget_s=20.0):
    import torch
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    one = time.perf_counter() - t0
    reps = max(1, min(reps, int(budget_s / max(one, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def measure(name, routes, reps, device="cuda:0", quiet=False):
    import torch
    s, model, X, Y = build(name, device)
    if not hasattr(model.kernel, "_auto_matrix_route"):
        model.kernel._auto_matrix_route = model.kernel.matrix_route
    out = dict(s)
    ref = None
    for route in routes:
        rec = {}
        try:
            set_route(model, route)
            feat = model.feature()

            def covs():
                return model.kernel.K_tens_n_seq_covs(model.Z, X, increments=True)

            def covs_fb():
                model.zero_grad(set_to_none=True)
                sum((a * a).sum() for a in covs()).backward()

            def step():
                model.zero_grad(set_to_none=True)
                (-model.elbo(X, Y)).backward()
            # the three level primitives on their own (scaled inputs): forward, forward + backward
            k = model.kernel
            with torch.no_grad():
                Xs0 = k.scale_sequences(k._seq3(X, False))
                Zs0 = k.scale_tensors(model.Z)
                fac0 = torch.ones((k.kern.num_levels + 1, Xs0.shape[0]), dtype=Xs0.dtype, device=Xs0.device)
            prims = {"kzz": lambda Z_, X_: k._tens_levels(Z_, True), "kzx": lambda Z_, X_: k._tvs_weighted(Z_, X_, fac0, True),
                     "kxx_diag": lambda Z_, X_: k._diag_levels(X_)}
            for pn, pf in prims.items():
                with torch.no_grad():
                    rec[pn + "_fwd_ms"] = timed(lambda: pf(Zs0, Xs0), reps)
                Zr, Xr = Zs0.clone().requires_grad_(True), Xs0.clone().requires_grad_(True)

                def fb():
                    Zr.grad = Xr.grad = None
                    o = pf(Zr, Xr)
                    (o * o).sum().backward()
                rec[pn + "_fwd_bwd_ms"] = timed(fb, reps)
            with torch.no_grad():
                vals = [a.detach().clone() for a in covs()]
                rec["covs_fwd_ms"] = timed(lambda: covs(), reps)
            rec["covs_fwd_bwd_ms"] = timed(covs_fb, reps)
            rec["step_ms"] = timed(step, reps)
            grads = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
            rec["finite"] = bool(torch.isfinite(grads).all()) and all(bool(torch.isfinite(v).all()) for v in vals)
            if ref is None:
                ref = (vals, grads.clone())
            else:
                rec["max_rel_diff_vs_first_route"] = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(vals, ref[0]))
                rec["grad_rel_diff_vs_first_route"] = float((grads - ref[1]).abs().max() / ref[1].abs().max())
            del feat
        except Exception as e:     # noqa: BLE001 -- a route that refuses a shape is a table entry, not a crash
            rec["error"] = "%s: %s" % (type(e).__name__, str(e)[:160])
        out[route] = rec
        if not quiet:
            print(json.dumps({"dataset": name, "route": route, **{k: v for k, v in s.items() if k != "name"}, **rec}), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="all")
    ap.add_argument("--routes", default="auto")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--json", default=None)
    ap.add_argument("--shapes", action="store_true", help="print the shapes and exit (no GPU needed)")
    a = ap.parse_args()
    names = sorted(DATASETS, key=lambda n: shape_of(n)["d_eff"]) if a.which == "all" else a.which.split(",")
    if a.shapes:
        for n in names:
            print(shape_of(n))
        return
    res = [measure(n, a.routes.split(","), a.reps) for n in names]
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
