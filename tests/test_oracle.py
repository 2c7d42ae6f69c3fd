"""Pins the CPU oracle (oracle/sigkern_oracle.py).

The reference holds no golden vectors; its only check is notebooks/signature_kernel.ipynb, three
identities against esig signature features on unseeded data (cells 6-29).  These tests re-run those
identities at the notebook's own shapes with an independent truncated-signature routine standing in
for esig, then pin the order-1 recursion by brute force, then check the committed fixtures are what
the oracle produces."""
import numpy as np
import pytest

from oracle import sigkern_oracle as O


def _relerr(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


@pytest.fixture(scope="module")
def notebook_data():
    rng = np.random.default_rng(4)
    M, N, L, d, T = 5, 100, 50, 3, 100         # notebook cells 4 and 15
    X = rng.standard_normal((N, L, d))
    Z = rng.standard_normal((M * (M + 1) // 2, T, d))
    sigs = np.stack([O.truncated_signature(x, M) for x in X])       # cell 6 (esig.tosig.stream2sig)
    tens = O.rank1_tensor_features(Z, M)                            # cell 18
    kern = O.SignatureKernelOracle(L * d, d, M, base="linear", order=M, normalization=False)  # cell 11
    Kl = kern.K(X.reshape(N, -1), return_levels=True)               # cell 11, levels kept apart
    return X.reshape(N, -1), Z, sigs, tens, kern, Kl


def test_notebook_identity_seq_vs_seq(notebook_data):
    X, Z, sigs, tens, kern, Kl = notebook_data
    K = Kl.sum(axis=0)                                              # cell 11 (variances = 1)
    K_sig = sigs @ sigs.T                                           # cell 8
    # the notebook reports Fro-norm 1.1e-8 on entries of size ~1e7 (cell 13)
    assert _relerr(K, K_sig) < 1e-12


def test_notebook_identity_levels(notebook_data):
    X, Z, sigs, tens, kern, Kl = notebook_data
    for m, sl in enumerate(O.signature_level_slices(3, 5)):
        assert _relerr(Kl[m], sigs[:, sl] @ sigs[:, sl].T) < 1e-12


def test_notebook_identity_tens_vs_seq(notebook_data):
    X, Z, sigs, tens, kern, Kl = notebook_data
    assert _relerr(kern.compute_K_tens_vs_seq(Z, X), tens @ sigs.T) < 1e-12   # cells 19-23


def test_notebook_identity_tens_vs_tens(notebook_data):
    X, Z, sigs, tens, kern, Kl = notebook_data
    assert _relerr(kern.compute_K_tens(Z), tens @ tens.T) < 1e-12             # cells 25-29


def test_baseline_config1_identity():
    """BASELINE.json configs[0]: N=64, L=32, d=3, num_levels=4, order=num_levels."""
    rng = np.random.default_rng(0)
    N, L, d, M = 64, 32, 3, 4
    X = rng.standard_normal((N, L, d))
    sigs = np.stack([O.truncated_signature(x, M) for x in X])
    kern = O.SignatureKernelOracle(L * d, d, M, base="linear", order=M, normalization=False)
    assert _relerr(kern.K(X.reshape(N, -1)), sigs @ sigs.T) < 1e-12


def test_first_order_is_strictly_increasing_tuple_sum():
    rng = np.random.default_rng(1)
    X = rng.standard_normal((3, 6, 2))
    Y = rng.standard_normal((2, 5, 2))
    kern = O.SignatureKernelOracle(12, 2, 3, base="linear", order=1, normalization=False)
    Kl = kern.K(X.reshape(3, -1), Y.reshape(2, -1), return_levels=True)
    dx, dy = np.diff(X, axis=1), np.diff(Y, axis=1)
    for i in range(3):
        for j in range(2):
            np.testing.assert_allclose(Kl[:, i, j], O.brute_force_first_order(dx[i] @ dy[j].T, 3), rtol=1e-12, atol=1e-13)


def test_higher_order_interpolates_between_first_order_and_signature():
    rng = np.random.default_rng(2)
    N, L, d, M = 5, 8, 2, 4
    X = rng.standard_normal((N, L, d)).reshape(N, -1)
    K = {o: O.SignatureKernelOracle(L * d, d, M, base="linear", order=o, normalization=False).K(X, return_levels=True)
         for o in (1, 2, 3, 4)}
    # levels <= order are exact signature levels; levels 0,1 agree for every order
    for o in (1, 2, 3):
        np.testing.assert_allclose(K[o][:o + 1], K[4][:o + 1], rtol=1e-11, atol=1e-12)
        assert _relerr(K[o][o + 1], K[4][o + 1]) > 1e-6
    # order clamp: order<=0 or >=M means M (kernels.py:57)
    assert O.SignatureKernelOracle(L * d, d, M, order=-1).order == M
    assert O.SignatureKernelOracle(L * d, d, M, order=7).order == M


def test_tens_vs_seq_higher_order_identity():
    rng = np.random.default_rng(3)
    N, L, d, M, T = 6, 9, 3, 4, 5
    X = rng.standard_normal((N, L, d))
    Z = rng.standard_normal((M * (M + 1) // 2, T, d))
    sigs = np.stack([O.truncated_signature(x, M) for x in X])
    tens = O.rank1_tensor_features(Z, M)
    kern = O.SignatureKernelOracle(L * d, d, M, base="linear", order=M, normalization=False)
    assert _relerr(kern.K_tens_vs_seq(Z, X.reshape(N, -1)), tens @ sigs.T) < 1e-12


def test_normalisation_quirks():
    """jitter goes on the diagonal before normalising; Kzz is never normalised."""
    rng = np.random.default_rng(5)
    N, L, d, M = 7, 6, 2, 3
    X = rng.standard_normal((N, L, d)).reshape(N, -1)
    kern = O.SignatureKernelOracle(L * d, d, M, base="rbf", normalization=True)
    Kl = kern.K(X, return_levels=True)
    np.testing.assert_allclose(np.diagonal(Kl, axis1=1, axis2=2), 1.0, rtol=1e-14)
    off = Kl[0][~np.eye(N, dtype=bool)]
    np.testing.assert_allclose(off, 1.0 / (1.0 + 1e-6), rtol=1e-14)          # level 0 off-diagonal
    np.testing.assert_allclose(kern.K(X, X, return_levels=True)[0], 1.0 / (1.0 + 1e-6), rtol=1e-14)
    np.testing.assert_allclose(kern.Kdiag(X), np.full(N, M + 1.0))
    Z = rng.standard_normal((M * (M + 1) // 2, 4, d))
    kern_nn = O.SignatureKernelOracle(L * d, d, M, base="rbf", normalization=False)
    np.testing.assert_allclose(kern.K_tens(Z), kern_nn.K_tens(Z))


def test_symmetric_equals_cross_without_normalisation():
    rng = np.random.default_rng(6)
    N, L, d, M = 6, 7, 3, 4
    X = rng.standard_normal((N, L, d)).reshape(N, -1)
    for base in O.BASE_KERNELS:
        bp = dict(alpha=rng.uniform(0.5, 1, 3), omega=0.3 * rng.standard_normal((3, d)), gamma=rng.uniform(0.5, 1.5, (3, d)), family="mixed") \
            if base == "spectral" else None
        kern = O.SignatureKernelOracle(L * d, d, M, base=base, normalization=False, base_params=bp)
        np.testing.assert_allclose(kern.K(X), kern.K(X, X), rtol=1e-12, atol=1e-12)
        Kd = kern.Kdiag(X, return_levels=True)
        np.testing.assert_allclose(Kd, np.diagonal(kern.K(X, return_levels=True), axis1=1, axis2=2), rtol=1e-10, atol=1e-12)


def test_constructor_validation():
    with pytest.raises(ValueError):
        O.SignatureKernelOracle(10, 3, 2)
    with pytest.raises(ValueError):
        O.SignatureKernelOracle(9, 3, 2, num_lags=-1)
    with pytest.raises(ValueError):
        O.SignatureKernelOracle(9, 3, 2, num_lags=1.5)


def test_lags_shapes_and_zero_lag_limit():
    rng = np.random.default_rng(7)
    X = rng.standard_normal((3, 9, 2))
    out = O.add_lags_to_sequences(X, np.array([0.125, 0.25]))
    assert out.shape == (3, 9, 3, 2)
    np.testing.assert_allclose(out[:, :, 0], X)
    # a lag of exactly one grid step (1/(L-1)) reproduces the sequence shifted by one, first value held
    np.testing.assert_allclose(out[:, 1:, 1], X[:, :-1], atol=1e-12)
    np.testing.assert_allclose(out[:, 0, 1], X[:, 0], atol=1e-12)


def test_committed_fixtures_match_oracle(golden):
    """The committed vectors are exactly what tests/golden/make_golden.py produces from the oracle."""
    cases, arr = golden
    names = {c["name"] for c in cases}
    assert {"nb_K", "nb_Kzx", "nb_Kzz", "c1_o1_n1", "c1_o4_n0"} <= names
    for c in cases:
        if c["method"] not in ("K", "K_tens", "K_tens_vs_seq"):
            continue
        kern = O.SignatureKernelOracle(**{k: (np.asarray(v) if isinstance(v, list) else v) for k, v in c["kern"].items()})
        n = c["name"]
        if c["method"] == "K":
            out = kern.K(arr[n + "/X"], arr[n + "/X2"] if "X2" in c["has"] else None, **c["call"])
        elif c["method"] == "K_tens":
            out = kern.K_tens(arr[n + "/Z"], **c["call"])
        else:
            out = kern.K_tens_vs_seq(arr[n + "/Z"], arr[n + "/X"], **c["call"])
        np.testing.assert_allclose(out, arr[n + "/out0"], rtol=1e-12, atol=1e-12, err_msg=n)


def test_fixture_notebook_cases_satisfy_signature_identity(golden):
    cases, arr = golden
    X = arr["nb_K/X"].reshape(40, 50, 3)
    sigs = np.stack([O.truncated_signature(x, 5) for x in X])
    tens = O.rank1_tensor_features(arr["nb_Kzx/Z"], 5)
    assert _relerr(arr["nb_K/out0"], sigs @ sigs.T) < 1e-12
    assert _relerr(arr["nb_Kzx/out0"], tens @ sigs.T) < 1e-12
    assert _relerr(arr["nb_Kzz/out0"], tens @ tens.T) < 1e-12


def test_svgp_algebra_restatement_properties():
    """oracle/svgp_oracle.py (GPflow 1.5.1 base_conditional / gauss_kl restated): closed-form sanity checks."""
    from oracle import svgp_oracle as SO
    rng = np.random.default_rng(8)
    M_, N_, R = 6, 9, 2
    A = rng.standard_normal((M_ + N_, M_ + N_ + 3))
    Kfull = A @ A.T + 1e-3 * np.eye(M_ + N_)
    Kmm, Kmn, Knn = Kfull[:M_, :M_], Kfull[:M_, M_:], Kfull[M_:, M_:]
    Lm = np.linalg.cholesky(Kmm)
    # KL of the prior against itself is zero, in both parametrisations
    assert abs(SO.gauss_kl(np.zeros((M_, R)), np.tile(np.eye(M_)[None], [R, 1, 1]))) < 1e-12
    assert abs(SO.gauss_kl(np.zeros((M_, R)), np.tile(Lm[None], [R, 1, 1]), K=Kmm)) < 1e-10
    assert abs(SO.gauss_kl(np.zeros((M_, R)), np.ones((M_, R)))) < 1e-12
    # q = prior (non-whitened: q_sqrt = chol Kmm) gives back the prior marginals
    f = np.zeros((M_, R))
    mean, var = SO.base_conditional(Kmn, Kmm, np.diag(Knn), f, q_sqrt=np.tile(Lm[None], [R, 1, 1]), white=False)
    np.testing.assert_allclose(mean, 0, atol=1e-12)
    np.testing.assert_allclose(var, np.tile(np.diag(Knn)[:, None], [1, R]), rtol=1e-9)
    mean, var = SO.base_conditional(Kmn, Kmm, Knn, f, full_cov=True, q_sqrt=np.tile(np.eye(M_)[None], [R, 1, 1]), white=True)
    np.testing.assert_allclose(var[0], Knn, rtol=1e-8, atol=1e-10)
    # exact GP posterior mean with a delta posterior at u = Kmm alpha
    u = rng.standard_normal((M_, R))
    mean, var = SO.base_conditional(Kmn, Kmm, np.diag(Knn), u, q_sqrt=None, white=False)
    np.testing.assert_allclose(mean, Kmn.T @ np.linalg.solve(Kmm, u), rtol=1e-8)
    np.testing.assert_allclose(var[:, 0], np.diag(Knn - Kmn.T @ np.linalg.solve(Kmm, Kmn)), rtol=1e-7, atol=1e-9)


def test_spectral_base_kernel_reduces_to_rbf_and_matern12():
    """kernels.py:921-942: with one component, omega = 0 and alpha = 1 the spectral kernel is the Gaussian kernel of the
    gamma-scaled points ('rbf' family) resp. exp(-r/2) ('exp' family); 'mixed' splits the components floor(Q/2) : rest."""
    rng = np.random.default_rng(9)
    X, Y = rng.standard_normal((7, 3)), rng.standard_normal((5, 3))
    g = rng.uniform(0.5, 1.5, (1, 3))
    one, zero = np.ones(1), np.zeros((1, 3))
    assert np.allclose(O.base_spectral(X, Y, one, zero, g, "rbf"), O.base_rbf(X * g, Y * g), rtol=1e-13)
    assert np.allclose(O.base_spectral(X, Y, one, zero, g, "exp"), np.exp(-O._euclid_dist(X * g, Y * g) / 2), rtol=1e-13)
    a, om, ga = rng.uniform(0.5, 1, 3), rng.standard_normal((3, 3)), rng.uniform(0.5, 1.5, (3, 3))
    mixed = O.base_spectral(X, Y, a, om, ga, "mixed")
    want = O.base_spectral(X, Y, a[:1], om[:1], ga[:1], "rbf") + O.base_spectral(X, Y, a[1:], om[1:], ga[1:], "exp")
    assert np.allclose(mixed, want, rtol=1e-13)
    # symmetric in its arguments, unit-free diagonal sum_q alpha_q
    assert np.allclose(np.diag(O.base_spectral(X, None, a, om, ga, "mixed")), a.sum())


@pytest.mark.parametrize("base", ["linear", "rbf"])
@pytest.mark.parametrize("difference", [True, False])
def test_c_restatement_equals_numpy_oracle(base, difference):
    """oracle/sigkern_ref.c (the C restatement timed as the all-cores CPU baseline of bench.py) against the NumPy oracle's level
    primitives, to rounding: sequence vs sequence (ragged lengths, every level) and tensor vs sequence with / without increments."""
    from oracle import cref
    rng = np.random.default_rng(8)
    M, d = 5, 3
    X, Y = rng.standard_normal((7, 12, d)) * 0.5, rng.standard_normal((5, 9, d)) * 0.5
    ko = O.SignatureKernelOracle(12 * d, d, M, base=base, difference=difference, lengthscales=None)
    for A, B in ((X, Y), (Y, X), (X, X)):
        got, want = cref.seq_levels(A, B, M, base, difference), ko._K_seq(A, B)
        assert got.shape == want.shape and np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    for incr in (False, True):
        Z = rng.standard_normal((M * (M + 1) // 2, 6, 2, d) if incr else (M * (M + 1) // 2, 6, d)) * 0.5
        got, want = cref.tens_vs_seq_levels(Z, X, M, base, difference), ko._K_tens_vs_seq(Z, X, increments=incr)
        assert got.shape == want.shape and np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    assert cref.threads() >= 1


def test_committed_low_rank_fixtures_match_oracle(golden_lowrank):
    """tests/golden/lowrank.npz is what the low-rank restatement gives for the committed random objects (landmarks, jitter draw,
    projections): a value pin of signature_algs.py:162-222 / low_rank_calculations.py:26-193 as restated, next to the statistical
    checks above."""
    cases, arr, sketches = golden_lowrank
    assert len(cases) >= 5
    for c in cases:
        n = c["name"]
        kw = {k: (np.asarray(v) if isinstance(v, list) else v) for k, v in c["kern"].items()}
        lo = O.LowRankOracle(O.SignatureKernelOracle(**kw), arr[f"{n}/landmarks"], arr[f"{n}/jitter_diag"], sketches(n, kw["num_levels"]))
        X, X2, Z, incr = arr[f"{n}/X"], arr[f"{n}/X2"], arr[f"{n}/Z"], c["increments"]
        got = dict(K=lo.K(X), Kx=lo.K(X, X2, return_levels=True), Kzx=lo.K_tens_vs_seq(Z, X, increments=incr), Kzz=lo.K_tens(Z, increments=incr))
        if "Kdiag" in c["outputs"]:
            got["Kdiag"] = lo.Kdiag(X)
        for k in c["outputs"]:
            want = arr[f"{n}/out/{k}"]
            assert np.abs(got[k] - want).max() <= 1e-11 * np.abs(want).max(), (n, k)


def test_higher_order_levels_of_a_one_column_sequence_are_better_conditioned_as_features():
    """Why a float64 outlier of the randomised sweep (profiles/r04_fuzz.txt, case 187: d = 1, 32 steps, order 3, normalised) is the ORACLE's: the
    pair recursion (signature_algs.py:37-74, restated op for op) sums products over index tuples that cancel by many orders of magnitude for a
    one-column sequence, the per-sequence truncated-exponential sweep (what the feature kernels run) does not.  Both against an 80-bit restatement
    of the features: the float64 features stay within 1e-10 of it after the normalisation, the recursion is orders of magnitude further away."""
    rng = np.random.default_rng(51)
    N, L, M, order = 40, 33, 5, 3
    X = np.cumsum(0.3 * rng.standard_normal((N, L, 1)), axis=1)

    def feats(x, dt):
        V = [dt(1)] + [dt(0)] * M
        for v in np.diff(np.asarray(x, dtype=dt)):
            new = list(V)
            for m in range(1, M + 1):
                acc, term = V[m], dt(1)
                for k in range(1, min(order, m) + 1):
                    term = term * v / dt(k)
                    acc = acc + V[m - k] * term
                new[m] = acc
            V = new
        return np.array(V, dtype=dt)

    def gram(dt):
        F = np.array([feats(x[:, 0], dt) for x in X])
        K = np.zeros((N, N), dtype=dt)
        for m in range(M + 1):
            a = F[:, m]
            s = np.sqrt(a * a + dt(1e-6))
            Km = np.outer(a, a) + dt(1e-6) * np.eye(N, dtype=dt)
            K += Km / np.outer(s, s)
        return K.astype(np.float64)

    ref, f64 = gram(np.longdouble), gram(np.float64)
    ko = O.SignatureKernelOracle(L, 1, M, base="linear", order=order, normalization=True, lengthscales=None)
    rec = ko.K(X.reshape(N, -1))
    scale = np.abs(ref).max()
    e_feat, e_rec = np.abs(f64 - ref).max() / scale, np.abs(rec - ref).max() / scale
    assert e_feat < 1e-10                                     # (9e-13 on this draw)
    assert 100.0 * e_feat < e_rec < 1e-2                      # (5e-8 on this draw, 1e-4 on the sweep's: the recursion's cancellation)


# ---- round 6: independent witness for the non-linear base kernels ------------------------------------------------------------
def _witness():
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(here, "witness.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(here, "witness.npz"))


def witness_kernel(make, c, normalization=False, variances=1):
    """make(input_dim, num_features, num_levels, **kw) -> kernel object (the oracle's class here, the product's in the GPU tests)."""
    kw = dict(lengthscales=c["lengthscales"], normalization=normalization, variances=variances,
              num_lags=(len(c["lags"]) if c["lags"] else None))
    return kw


def _oracle_for(c, L, normalization=False, variances=1):
    k = O.SignatureKernelOracle(L * c["d"], c["d"], c["M"], base=c["base"], base_params=c["params"],
                                **witness_kernel(None, c, normalization, variances))
    if c["lags"]:
        k.lags, k.gamma = np.asarray(c["lags"], dtype=float), np.asarray(c["gamma"], dtype=float)
    return k


def _lvl_err(got, want):
    """max over levels of the matrix-relative error."""
    return max(float(np.abs(g - w).max() / np.abs(w).max()) for g, w in zip(got, want))


def test_oracle_against_independent_witness_values():
    """tests/golden/make_witness.py: kappa from closed forms in 50-digit arithmetic (no oracle.base_*), levels as literal sums over increasing
    index tuples.  rbf, matern12/32/52, poly, mix, cosine; with and without lengthscales and lags; seq x seq (cross, symmetric, normalised),
    tensor x seq and tensor x tensor with and without increments.  The oracle stays within 1e-10 of every value."""
    meta, W = _witness()
    assert {c["base"] for c in meta} == {"rbf", "matern12", "matern32", "matern52", "poly", "mix", "cosine"}
    for c in meta:
        n = c["name"]
        X, Y, Z, Zi = (W[n + "/" + k] for k in ("X", "Y", "Z", "Zi"))
        nx, L1, d = X.shape
        ny, L2, _ = Y.shape
        k1 = _oracle_for(c, L1)
        cross = k1.K(X.reshape(nx, -1), Y.reshape(ny, -1), return_levels=True) if L1 == L2 else None
        if cross is None:                       # the class takes one length per object (input_dim): the levels come from _K_seq on scaled inputs
            cross = k1._K_seq(k1._apply_scaling_and_lags_to_sequences(X), _oracle_for(c, L2)._apply_scaling_and_lags_to_sequences(Y))
        assert _lvl_err(cross, W[n + "/K_cross_levels"]) < 1e-10, n
        sym = k1.K(X.reshape(nx, -1), return_levels=True)
        want = W[n + "/K_symm_levels"]
        off = ~np.eye(nx, dtype=bool)
        if c["base"] == "matern12":
            # a sequence against itself: the reference's float64 squared distance of coinciding points is rounding noise, its root 1e-8 -- the
            # closed form has exactly 0 there; off-diagonal pairs are held as everything else
            assert max(float(np.abs(g[off] - w[off]).max() / np.abs(w).max()) for g, w in zip(sym, want)) < 1e-10, n
        else:
            assert _lvl_err(sym, want) < 1e-10, n
            kn = _oracle_for(c, L1, normalization=True, variances=W[n + "/variances"])
            assert float(np.abs(kn.K(X.reshape(nx, -1)) - W[n + "/K_symm_normalised"]).max()) < 1e-10, n
        for tag, ZZ, inc in (("", Z, False), ("_incr", Zi, True)):
            assert _lvl_err(k1.K_tens_vs_seq(ZZ, X.reshape(nx, -1), return_levels=True, increments=inc), W[n + "/Kzx%s_levels" % tag]) < 1e-10, (n, tag)
            kzz, wzz = k1.K_tens(ZZ, return_levels=True, increments=inc), W[n + "/Kzz%s_levels" % tag]
            if c["base"] == "matern12":         # a tensor against itself: coinciding points again
                offz = ~np.eye(kzz.shape[1], dtype=bool)
                kzz, wzz = kzz[:, offz], wzz[:, offz]
            assert _lvl_err(kzz, wzz) < 1e-10, (n, tag)
