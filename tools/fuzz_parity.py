"""Randomised parity sweep of the evaluation path against the CPU oracle (run on a GPU box):
    python tools/fuzz_parity.py [cases] [seed]
Draws kernel families, orders, flags, dtypes and ragged shapes at random and reports every case whose relative error exceeds
the tolerance (float64 1e-6, float32 1e-4 on the matrix scale)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

BASES = ["linear", "rbf", "cosine", "poly", "mix", "matern12", "matern32", "matern52"]


def classes():
    from gpsig_amd import kernels as K
    return {"linear": K.SignatureLinear, "rbf": K.SignatureRBF, "cosine": K.SignatureCosine, "poly": K.SignaturePoly, "mix": K.SignatureMix,
            "matern12": K.SignatureMatern12, "matern32": K.SignatureMatern32, "matern52": K.SignatureMatern52}


def relerr(got, want, f32):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want)
    if f32:
        return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-300))
    return float(np.max(np.abs(got - want) / (np.abs(want) + 1e-6 * np.abs(want).max() + 1e-300)))


def draw_case(rng):
    """One case of the sweep: every draw the sweep makes for it, in the sweep's order, and nothing that needs a GPU -- so that
    tests/golden/make_fuzz_cases.py can replay a seed to the case a sweep reported (committed fixtures: tests/golden/fuzz_cases.npz)."""
    base = rng.choice(BASES)
    M = int(rng.integers(1, 7))
    order = int(rng.choice([1, 1, 1, 2, 3, M]))
    # FUZZ_WIDE=1 (round 6): the widths of the reference's own run settings, where the wide route (csrc/wide_api.hip) takes Kzx / Kzz / the lattices
    d = int(rng.choice([9, 12, 14, 23, 33, 50, 63, 130] if os.environ.get("FUZZ_WIDE") else [1, 2, 3, 5, 8, 11, 16, 20]))
    lags = int(rng.choice([0, 0, 0, 1, 2])) if base != "poly" else 0
    L1, L2 = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 100, 130])), int(rng.choice([1, 2, 5, 17, 32, 65, 90]))
    N1, N2 = int(rng.integers(1, 40)), int(rng.integers(1, 20))
    if base in ("linear", "cosine") and rng.integers(0, 3) == 0:          # sizes that cross the contraction's 128-wide tiles and its depth pieces
        N1, N2 = int(rng.choice([127, 129, 200, 300])), int(rng.choice([1, 64, 130, 260]))
        L1, L2 = min(L1, 33), min(L2, 32)
    norm, diff = bool(rng.integers(0, 2)), bool(rng.integers(0, 4) > 0)
    f32 = bool(rng.integers(0, 5) == 0)
    if (L1 == 1 or L2 == 1) and diff:
        L1, L2 = max(L1, 2), max(L2, 2)
    if lags and min(L1, L2) < 3:
        lags = 0
    kw = dict(num_levels=M, order=order, normalization=norm, difference=diff, num_lags=lags or None,
              lengthscales=rng.uniform(0.7, 1.6, d), variances=rng.uniform(0.5, 1.5, M + 1))
    desc = dict(base=str(base), M=M, order=order, d=d, lags=lags, L1=L1, L2=L2, N1=N1, N2=N2, norm=norm, diff=diff, f32=f32)
    scale = 0.3 / np.sqrt(d)
    X = np.cumsum(scale * rng.standard_normal((N1, L1, d)), axis=1).reshape(N1, -1)
    X2 = np.cumsum(scale * rng.standard_normal((N2, L2, d)), axis=1).reshape(N2, -1)
    if base == "cosine":
        X, X2 = X + 1.0, X2 + 1.0
    lt = M * (M + 1) // 2
    incr = bool(rng.integers(0, 2))
    T = int(rng.integers(1, 9)) if rng.integers(0, 4) else int(rng.choice([33, 64, 70, 130]))
    # library knobs that route a case through the round-2 kernels whatever its size: the Kzx tile kernel below 32 tensors,
    # the packed float32 kernels with one or four waves per ring, for the linear family too
    opts = dict(tvs_tile=int(rng.choice([-1, 1])), f32_waves=int(rng.choice([0, 1, 4])), pk2=int(rng.choice([1, 2])),
                diag_own=int(rng.choice([1, 1, 0])), tens_tile=int(rng.choice([1, 1, 0])), lr_fused=int(rng.choice([1, 1, 0])),
                lr_gemm=int(rng.choice([1, 1, 0])),
                # round 3: SignatureLinear's Gram as a contraction of explicit level features wherever it is built (1), by the
                # planner's choice (-1), never (0); its two contraction kernels
                sig_features=int(rng.choice([1, 1, -1, 0])), sig_gemm_dma=int(rng.choice([1, 1, 0])))
    de = d * (lags + 1)
    Z = 0.5 * rng.standard_normal((lt, T, 2, de) if incr else (lt, T, de)) + (1.0 if base == "cosine" else 0.0)
    dt = np.float32 if f32 else np.float64
    Xq, X2q, Zq = X.astype(dt), X2.astype(dt), Z.astype(dt)
    case = dict(base=str(base), M=M, order=order, d=d, lags=lags, L1=L1, L2=L2, N1=N1, N2=N2, norm=norm, diff=diff, f32=f32, kw=kw, desc=desc, opts=opts,
                incr=incr, T=T, de=de, Xq=Xq, X2q=X2q, Zq=Zq, lowrank=None)
    # low-rank mode (order 1, float64): drawn here so that the stream of a seed does not depend on what the evaluation does
    if order == 1 and not f32 and base != "cosine" and rng.integers(0, 3) == 0 and min(L1, L2) >= 2:
        c_ = int(rng.choice([3, 8, 17, 50]))
        r_ = int(rng.choice([2, 9, 30, 50]))
        sp = str(rng.choice(["sqrt", "log", "lin"]))
        npts = N1 * L1 + Zq.reshape(-1, de).shape[0]          # the landmarks are drawn from X and Z
        c_ = min(c_, npts)
        if sp == "lin":
            r_ = min(r_, c_ * min(c_, r_))
        case["lowrank"] = dict(c=c_, r=r_, sparsity=sp, seed=int(rng.integers(1 << 30)))
    return case


def oracle_for(case, dtype=np.float64):
    from oracle import sigkern_oracle as O
    ko = O.SignatureKernelOracle(case["L1"] * case["d"], case["d"], base=case["base"], dtype=dtype, **case["kw"],
                                 base_params=({"gamma": 1.0, "degree": 3.0} if case["base"] == "poly" else None))
    ko.input_dim = case["L1"] * case["d"]
    return ko


def main():
    from gpsig_amd import _lib
    from oracle import sigkern_oracle as O
    CTX = _lib.context(0, 0)
    CLASS = classes()
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for it in range(cases):
        desc = None
        try:
            cs = draw_case(rng)
            base, M, order, d, L1, L2, f32, kw, desc, incr, T, de = (cs[k] for k in ("base", "M", "order", "d", "L1", "L2", "f32", "kw", "desc", "incr", "T", "de"))
            for k_, v_ in cs["opts"].items():
                CTX.set_option(k_, v_)
            if os.environ.get("FUZZ_WIDE"):
                CTX.set_option("wide", 1 if it % 3 == 0 else -1)
                CTX.set_option("wide_chunk_mb", 1 if it % 5 == 0 else 0)
            desc = dict(desc, **cs["opts"])
            kx1 = CLASS[base](L1 * d, d, **kw)
            ko1 = oracle_for(cs)
            Xq, X2q, Zq = cs["Xq"], cs["X2q"], cs["Zq"]
            Xo, X2o, Zo = Xq.astype(np.float64), X2q.astype(np.float64), Zq.astype(np.float64)
            tol = 1e-4 if f32 else 1e-6
            checks = [("K", lambda: kx1.K(Xq), lambda: ko1.K(Xo)), ("Kdiag", lambda: kx1.Kdiag(Xq), lambda: ko1.Kdiag(Xo)),
                      ("Kzx", lambda: kx1.K_tens_vs_seq(Zq, Xq, increments=incr), lambda: ko1.K_tens_vs_seq(Zo, Xo, increments=incr)),
                      ("Kzz", lambda: kx1.K_tens(Zq, increments=incr), lambda: ko1.K_tens(Zo, increments=incr)),
                      ("Kx", lambda: kx1.K(Xq, X2q, presliced=True), lambda: ko1.K(Xo, X2o) if L1 == L2 else _cross(ko1, Xo, X2o, d))]
            for name, g, w in checks:
                try:
                    got = g()
                except NotImplementedError as e:
                    print(f"[{it}] {name}: NotImplementedError ({str(e)[:90]}) {desc}")
                    continue
                err = relerr(got, w(), f32)
                if not (err <= tol):
                    bad += 1
                    print(f"[{it}] {name}: rel.err {err:.3e} > {tol}  {desc} incr={incr} T={T}")
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.savez(f"gpurun_out/fuzz_fail_{it}_{name}.npz", X=Xo, X2=X2o, Z=Zo, got=np.asarray(got, dtype=np.float64), want=w(),
                             ls=kw["lengthscales"], var=kw["variances"], desc=str(desc), incr=incr)
            # low-rank mode (order 1, float64): the product against the oracle's restatement GIVEN THE SAME random objects; judged on
            # the scale of each array (two correct eigensolvers differ in the near-null space of an ill-conditioned landmark Gram)
            if cs["lowrank"]:
                lr = cs["lowrank"]
                kxl = CLASS[base](L1 * d, d, low_rank=True, num_components=lr["c"], rank_bound=lr["r"], sparsity=lr["sparsity"], **kw)
                kxl.rng = np.random.default_rng(lr["seed"])
                st = kxl.draw_low_rank(X=Xo, Z=Zo, increments=incr)
                lo = O.LowRankOracle(ko1, st.landmarks, st.jitter_diag, st.sketches)
                ldesc = dict(desc, c=lr["c"], r=lr["r"], sparsity=lr["sparsity"])
                for name, g, w in (("lrK", lambda: kxl.K(Xo, lr_state=st), lambda: lo.K(Xo)),
                                   ("lrKzx", lambda: kxl.K_tens_vs_seq(Zo, Xo, increments=incr, lr_state=st), lambda: lo.K_tens_vs_seq(Zo, Xo, increments=incr)),
                                   ("lrKzz", lambda: kxl.K_tens(Zo, increments=incr, lr_state=st), lambda: lo.K_tens(Zo, increments=incr))):
                    got, want = np.asarray(g()), w()
                    err = float(np.abs(got - want).max() / (np.abs(want).max() + 1e-300))
                    if not (err <= 1e-4):
                        bad += 1
                        print(f"[{it}] {name}: rel.err {err:.3e} > 1e-4  {ldesc} incr={incr} T={T}")
        except Exception:
            bad += 1
            print(f"[{it}] EXCEPTION {desc}")
            traceback.print_exc()
    print(f"fuzz: {cases} cases, {bad} failures")


def _cross(ko, X, X2, d):
    # the oracle class reshapes by num_features, so ragged lengths just work
    return ko.K(X, X2)


if __name__ == "__main__":
    main()
