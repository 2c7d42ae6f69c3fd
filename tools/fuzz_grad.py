"""Randomised sweep of the gradient path against torch.autograd of the differentiable oracle (run on a GPU box):
    python tools/fuzz_grad.py [cases] [seed]
FUZZ_ORDER=1 also draws the algorithm's order (1 .. num_levels); GPSIG_OPTIONS="sig_features_grad=1" sends the linear / cosine kernel's
K(X), K(X, X2) through the feature space (the one-op level sum, gpsig_kernel_K_grad) at these small sizes too; FUZZ_FEATURES=1 does the same for
the training path's level-feature route of Kzx and the level diagonals (SignatureKernelModule.feature_route = "always")."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsig_amd import kernels as K, autodiff
from oracle import sigkern_oracle_torch as OT

CLASS = {"linear": K.SignatureLinear, "rbf": K.SignatureRBF, "cosine": K.SignatureCosine, "poly": K.SignaturePoly, "mix": K.SignatureMix,
         "matern12": K.SignatureMatern12, "matern32": K.SignatureMatern32, "matern52": K.SignatureMatern52}


def rel(a, b):
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-9))        # gradients that vanish identically (scale-free kernels) are noise


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    bad = 0
    for it in range(cases):
        base = str(rng.choice(list(CLASS)))
        M = int(rng.integers(1, 7))
        d = int(rng.choice([9, 12, 14, 23, 33, 70] if os.environ.get("FUZZ_WIDE") else [1, 2, 3, 5, 8, 11]))
        if base == "cosine" and d == 1:
            d = 2       # the cosine of two scalars is +-1 and the kernel ignores the one lengthscale: gradients that vanish identically, on both sides
                        # rounding noise -- 21 of the 22 entries above tolerance in round 5's sweeps (profiles/r05_fuzz_grad.txt) were of this class
        lags = int(rng.choice([0, 0, 1, 2])) if base != "poly" else 0
        L1, L2 = int(rng.choice([3, 4, 9, 20, 40, 70])), int(rng.choice([3, 5, 12, 33]))
        N1, N2, T = int(rng.integers(1, 9)), int(rng.integers(1, 7)), int(rng.integers(1, 7))
        norm, diff, incr = bool(rng.integers(0, 2)), bool(rng.integers(0, 4) > 0), bool(rng.integers(0, 2))
        order = int(rng.integers(1, M + 1)) if os.environ.get("FUZZ_ORDER") and rng.integers(0, 2) else 1
        desc = dict(base=base, M=M, d=d, lags=lags, L1=L1, L2=L2, N1=N1, N2=N2, T=T, norm=norm, diff=diff, incr=incr, order=order)
        try:
            kern = CLASS[base](max(L1, L2) * d, d, M, normalization=norm, difference=diff, num_lags=lags or None, order=order,
                               lengthscales=rng.uniform(0.8, 1.6, d), variances=rng.uniform(0.5, 1.5, M + 1))
            mod = autodiff.SignatureKernelModule(kern, device=dev)
            if os.environ.get("FUZZ_FEATURES"):
                mod.feature_route = "always"        # the linear / cosine kernel's Kzx and level diagonals through the level features at these sizes too
            leaf = lambda t: None if t is None else t.detach().cpu().clone().requires_grad_(True)
            orc = OT.SignatureKernelTorchOracle(d, M, base, variances=leaf(mod.variances), sigma=leaf(mod.sigma), lengthscales=leaf(mod.lengthscales),
                                                normalization=norm, difference=diff, num_lags=lags, lags=leaf(mod.lags) if lags else None,
                                                gamma=leaf(mod.gamma) if lags else None, p0=leaf(mod.p0), p1=kern._current_base_params()[1], order=order)
            sc = 0.4 / np.sqrt(d)
            off = 1.0 if base == "cosine" else 0.0
            X = np.cumsum(sc * rng.standard_normal((N1, L1, d)), axis=1).reshape(N1, -1) + off
            X2 = np.cumsum(sc * rng.standard_normal((N2, L2, d)), axis=1).reshape(N2, -1) + off
            de, lt = d * (lags + 1), M * (M + 1) // 2
            Z = 0.5 * rng.standard_normal((lt, T, 2, de) if incr else (lt, T, de)) + off
            W = [rng.standard_normal(s) for s in ((T, T), (T, N1), (N1,), (N1, N1), (N1, N2))]
            def loss(m, Zt, Xt, X2t, cv):
                Kzz, Kzx, Kxx = m.K_tens_n_seq_covs(Zt, Xt, increments=incr)
                return (Kzz * cv(W[0])).sum() + (Kzx * cv(W[1])).sum() + (Kxx * cv(W[2])).sum() + (m.K(Xt) * cv(W[3])).sum() + (m.K(Xt, X2t) * cv(W[4])).sum()
            Zg, Xg, X2g = (torch.tensor(a, device=dev, requires_grad=True) for a in (Z, X, X2))
            lg = loss(mod, Zg, Xg, X2g, lambda a: torch.tensor(a, device=dev))
            lg.backward()
            Zc, Xc, X2c = (torch.tensor(a, requires_grad=True) for a in (Z, X, X2))
            lc = loss(orc, Zc, Xc, X2c, torch.tensor)
            lc.backward()
            errs = {"loss": abs(lg.item() - lc.item()) / max(1.0, abs(lc.item())), "Z": rel(Zg.grad, Zc.grad), "X": rel(Xg.grad, Xc.grad), "X2": rel(X2g.grad, X2c.grad)}
            sig = lambda r: torch.sigmoid(r.detach().cpu())
            errs["var"] = rel(mod.raw_variances.grad, orc.variances.grad * sig(mod.raw_variances))
            errs["ls"] = rel(mod.raw_lengthscales.grad, orc.lengthscales.grad * sig(mod.raw_lengthscales))
            if lags:
                r = mod.raw_lags.detach().cpu()
                errs["lags"] = rel(mod.raw_lags.grad, orc.lags.grad * torch.sigmoid(r) * (1 - torch.sigmoid(r)))
            if mod.raw_p0 is not None:
                errs["p0"] = rel(mod.raw_p0.grad, orc.p0.grad * sig(mod.raw_p0))
            worst = max(errs.values())
            if not (worst < 1e-6):
                bad += 1
                print(f"[{it}] worst {worst:.2e} {errs} {desc}")
        except Exception:
            bad += 1
            print(f"[{it}] EXCEPTION {desc}")
            traceback.print_exc()
    print(f"fuzz_grad: {cases} cases, {bad} failures")


if __name__ == "__main__":
    main()
