"""CPU checks of the seq-gram kernel's dataflow through the lock-step wave emulator (tests/emu/):
the per-lane code, pair hand-over, LDS-ring reuse, task coverage, config selection and the on-chip
epilogue are the SAME C++ the gfx950 kernel compiles (gpsig_amd/csrc/seq_*.hpp); only DPP/LDS/global
memory are emulated.  The GPU parity tests proper are in test_gpu_parity.py."""
import numpy as np
import pytest

import emu_util as E
from oracle import sigkern_oracle as O


def rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


def walks(rng, N, L, d, scale=0.3):
    return np.cumsum(scale * rng.standard_normal((N, L, d)), axis=1)


def oracle(base, L, d, M, difference=True, bp=None):
    return O.SignatureKernelOracle(L * d, d, M, base=base, normalization=False, lengthscales=None,
                                   difference=difference, base_params=bp or {})


@pytest.mark.parametrize("N,L,d,M", [(9, 13, 3, 4), (5, 32, 3, 4), (6, 64, 8, 5), (3, 50, 3, 5), (4, 100, 2, 4),
                                     (7, 17, 4, 8), (4, 2, 1, 1), (3, 3, 2, 6), (70, 5, 2, 3)])
def test_linear_symmetric_levels(N, L, d, M):
    rng = np.random.default_rng(N * 1000 + L)
    X = rng.standard_normal((N, L, d))
    out, cfg = E.seq_levels(X, None, "linear", M)
    assert not np.isnan(out).any(), "circulant task list must cover every entry (with the mirror)"
    assert rel(out, oracle("linear", L, d, M)._K_seq(X)) < 1e-12
    np.testing.assert_array_equal(out, out.transpose(0, 2, 1))


@pytest.mark.parametrize("base,bp", [("rbf", {}), ("cosine", {}), ("poly", dict(gamma=0.7, degree=3.0)),
                                     ("mix", dict(mixing=0.3)), ("matern12", {}), ("matern32", {}), ("matern52", {})])
def test_point_kernels_cross_and_symmetric(base, bp):
    rng = np.random.default_rng(11)
    X, Y = walks(rng, 7, 21, 4), walks(rng, 6, 13, 4)
    k = oracle(base, 21, 4, 4, bp=bp)
    p = (bp.get("gamma", bp.get("mixing", 0.0)), bp.get("degree", 0.0))
    # Matern-1/2 takes sqrt(max(r^2, 1e-40)) of a squared distance that is pure rounding noise when a
    # point meets itself (kernels.py:779-781): kappa(x, x) = exp(-sqrt(noise)) is only reproducible to
    # ~1e-8, in the reference as much as here.  Everything else agrees to rounding.
    tol = 1e-6 if base == "matern12" else 1e-11
    out, _ = E.seq_levels(X, Y, base, 4, base_params=p)
    assert rel(out, k._K_seq(X, Y)) < 1e-11
    out, _ = E.seq_levels(X, None, base, 4, base_params=p)
    assert rel(out, k._K_seq(X)) < tol
    out, _ = E.seq_levels(X, None, base, 4, base_params=p, diag_only=True)
    assert rel(out, k._K_seq_diag(X)) < tol


@pytest.mark.parametrize("base", ["linear", "rbf"])
def test_no_difference(base):
    rng = np.random.default_rng(12)
    X, Y = 0.4 * walks(rng, 5, 9, 3), 0.4 * walks(rng, 4, 12, 3)
    k = oracle(base, 9, 3, 3, difference=False)
    out, _ = E.seq_levels(X, Y, base, 3, difference=False)
    assert rel(out, k._K_seq(X, Y)) < 1e-12
    out, _ = E.seq_levels(X, None, base, 3, difference=False)
    assert rel(out, k._K_seq(X)) < 1e-12


@pytest.mark.parametrize("L1,L2", [(3, 40), (40, 3), (2, 2), (17, 64), (120, 9)])
def test_ragged_lengths_and_short_runs(L1, L2):
    """x side shorter than the 16-lane skew (ring deeper than 3), y side of any length that fits."""
    rng = np.random.default_rng(L1 * 100 + L2)
    X, Y = walks(rng, 37, L1, 2), walks(rng, 5, L2, 2)
    for base in ("linear", "rbf"):
        out, _ = E.seq_levels(X, Y, base, 3)
        assert rel(out, oracle(base, L1, 2, 3)._K_seq(X, Y)) < 1e-12


def test_exact_and_generic_variants_agree_bitwise():
    rng = np.random.default_rng(3)
    X = rng.standard_normal((6, 30, 3))
    a, ca = E.seq_levels(X, None, "linear", 4, allow_exact=True)
    b, cb = E.seq_levels(X, None, "linear", 4, allow_exact=False)
    assert ca["exact"] == 1 and cb["exact"] == 0
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 8, 9, 16, 17])
def test_circulant_cover_every_pair_exactly_once(N):
    """PRED_CIRCULANT: every unordered pair is emitted exactly once, for even and odd N."""
    rng = np.random.default_rng(N)
    X = rng.standard_normal((N, 6, 2))
    out, _ = E.seq_levels(X, None, "linear", 2)
    assert not np.isnan(out).any()
    assert rel(out, oracle("linear", 6, 2, 2)._K_seq(X)) < 1e-13
    # emission count: run the same tasks without the mirror; exactly N(N+1)/2 entries get written
    gy = E.geometry("linear", True, 6, 4)
    cfg = E.select(gy["rows"], 2, 2)
    g = E.geometry("linear", True, 6, cfg["D"])
    rec = E.build_records(X, g, True, cfg["D"])
    o = np.full((3, N, N), np.nan)
    E.run(cfg, g, g, rec, rec, N, N, 2, 0, 0.0, 0.0, o, N, 1, N * N, None, None, 0.0, False, E.PRED_CIRCULANT, False)
    written = ~np.isnan(o[1])
    assert written.sum() == N * (N + 1) // 2
    assert not (written & written.T & ~np.eye(N, dtype=bool)).any()


def test_shards_partition_the_work():
    rng = np.random.default_rng(5)
    X = rng.standard_normal((23, 9, 2))
    ref = oracle("linear", 9, 2, 3)._K_seq(X)
    acc = np.zeros_like(ref)
    seen = np.zeros(ref.shape, dtype=int)
    for r in range(3):
        o, _ = E.seq_levels(X, None, "linear", 3, max_run=5, shard=(r, 3))
        m = ~np.isnan(o)
        acc[m] = o[m]
        seen += m
    # shards are disjoint up to the mirror image of the same pair, and together complete
    assert (seen >= 1).all()
    assert rel(acc, ref) < 1e-13


@pytest.mark.parametrize("base,norm", [("linear", True), ("linear", False), ("rbf", True), ("rbf", False)])
def test_on_chip_epilogue_matches_kernel_K(base, norm):
    """diag pass -> factors -> normalise / weight / sum at the pair boundary == SignatureKernel.K."""
    rng = np.random.default_rng(21)
    L, d, M = 19, 3, 4
    X, Y = walks(rng, 10, L, d), walks(rng, 6, L, d)
    var = 0.5 + rng.random(M + 1)
    k = O.SignatureKernelOracle(L * d, d, M, base=base, normalization=norm, lengthscales=None, variances=var)
    k.sigma = 1.7
    for X2 in (None, Y):
        for lv in (False, True):
            got = E.kernel_K(X, X2, base, M, var, 1.7, norm, return_levels=lv)
            want = k.K(X.reshape(10, -1), None if X2 is None else X2.reshape(6, -1), return_levels=lv)
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)


def test_golden_fixture_cases_through_emulator(golden):
    cases, arr = golden
    done = 0
    for c in cases:
        kw = c["kern"]
        if c["method"] != "K" or kw.get("order", 1) != 1 or kw.get("num_lags") or kw["num_levels"] > 8:
            continue
        n = c["name"]
        d, M = kw["num_features"], kw["num_levels"]
        ls = kw.get("lengthscales", 1)
        scale = np.ones(d) if ls is None else np.asarray(ls) * np.ones(d)
        X = arr[n + "/X"].reshape(arr[n + "/X"].shape[0], -1, d) / scale
        X2 = arr[n + "/X2"].reshape(arr[n + "/X2"].shape[0], -1, d) / scale if "X2" in c["has"] else None
        if X.shape[1] > 64 or X.shape[0] > 30:
            continue  # keep the CPU suite quick; the GPU parity test runs every case
        bp = kw.get("base_params") or {}
        p = (bp.get("gamma", bp.get("mixing", 0.0)), bp.get("degree", 0.0))
        try:
            got = E.kernel_K(X, X2, kw["base"], M, np.asarray(kw.get("variances", 1)) * np.ones(M + 1), 1.0,
                             kw.get("normalization", True), kw.get("difference", True), p,
                             return_levels=c["call"].get("return_levels", False))
        except NotImplementedError:
            continue  # shape outside the emulator's small config table (the product table is larger)
        tol = 1e-6 if kw["base"] == "matern12" else 1e-10   # see test_point_kernels_cross_and_symmetric
        np.testing.assert_allclose(got, arr[n + "/out0"], rtol=tol, atol=1e-12, err_msg=n)
        done += 1
    assert done >= 30


# ------------------------------------------------------------------------------------------------
# higher-order algorithm (signature_algs.py:37-74) through the emulated kernel
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", E.HO_TABLE, ids=lambda c: f"G{c['G']}C{c['C']}o{c['OMAX']}")
@pytest.mark.parametrize("N,L,d,M,order", [(5, 9, 3, 4, 2), (5, 9, 3, 4, 3), (4, 12, 2, 4, 4), (3, 20, 2, 5, 2),
                                           (3, 9, 2, 2, 2), (3, 9, 2, 1, 1), (4, 8, 2, 4, 1), (3, 30, 3, 6, 6)])
def test_higher_order_levels(cfg, N, L, d, M, order):
    if M > cfg["MMAX"] or order > cfg["OMAX"] or L > cfg["G"] * cfg["C"]:
        pytest.skip("outside this emulator shape")
    rng = np.random.default_rng(N + L + M + order)
    X, Y = rng.standard_normal((N, L, d)), rng.standard_normal((3, max(L - 2, 2), d))
    for base in ("linear", "rbf"):
        k = O.SignatureKernelOracle(L * d, d, M, base=base, normalization=False, lengthscales=None, order=order)
        assert rel(E.seq_levels_ho(X, None, base, M, order, cfg), k._K_seq(X)) < 1e-12
        assert rel(E.seq_levels_ho(X, Y, base, M, order, cfg), k._K_seq(X, Y)) < 1e-12
        assert rel(E.seq_levels_ho(X, None, base, M, order, cfg, diag_only=True), k._K_seq_diag(X)) < 1e-12


def test_full_order_kernel_is_the_signature_inner_product():
    """notebooks/signature_kernel.ipynb cells 6-13 (order = num_levels vs signature features), through the emulator."""
    rng = np.random.default_rng(77)
    N, L, d, M = 6, 14, 3, 4
    X = rng.standard_normal((N, L, d))
    sigs = np.stack([O.truncated_signature(x, M) for x in X])
    got = E.seq_levels_ho(X, None, "linear", M, M, E.HO_TABLE[0])
    for m, sl in enumerate(O.signature_level_slices(d, M)):
        assert rel(got[m], sigs[:, sl] @ sigs[:, sl].T) < 1e-12
