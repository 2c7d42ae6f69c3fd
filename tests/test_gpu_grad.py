"""GPU parity of the gradient path (through the C ABI) against torch.autograd of the differentiable oracle.

The reference has no gradient code and no gradient tests: it trains by TensorFlow autodiff of the graph that
oracle/sigkern_oracle_torch.py restates (pinned on CPU by tests/test_grad_core.py).  Tolerance: float64, 1e-6 relative
to the largest gradient entry (north_star's tolerance for the kernel entries); observed ~1e-13.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import sigkern_oracle_torch as OT
from oracle import sigkern_oracle as O

pytestmark = pytest.mark.gpu

BASES = ["linear", "rbf", "cosine", "poly", "mix", "matern12", "matern32", "matern52"]
_P = C.POINTER(C.c_double)


def _bp(base):
    return (1.0, 3.0) if base == "poly" else ((0.4, 0.0) if base == "mix" else (0.0, 0.0))


def _t_kern(base, d, M, **kw):
    p0, p1 = _bp(base)
    return OT.SignatureKernelTorchOracle(d, M, base, p0=torch.tensor(p0, dtype=torch.float64, requires_grad=True), p1=p1, **kw)


def rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def _host_ctx():
    from gpsig_amd import _lib
    ctx = _lib.context(0, 0)
    ctx.set_pointer_mode(_lib.PTR_HOST)
    return ctx


def _params(base, d, M, difference, keep, order=1):
    from gpsig_amd.autodiff import _Spec
    p0, p1 = _bp(base)
    return _Spec(base, M, difference, p1, order=order).params(d, p0, keep)


def _vp(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


@pytest.mark.parametrize("base", BASES)
@pytest.mark.parametrize("difference", [True, False])
def test_seq_level_gradients(base, difference):
    rng = np.random.default_rng(21)
    ctx = _host_ctx()
    tol = 1e-6
    for (M, N1, N2, L1, L2, d, kind) in [(4, 70, 5, 9, 7, 3, "cross"), (3, 9, 9, 6, 6, 2, "sym"), (5, 130, 130, 8, 8, 5, "diag"), (2, 3, 4, 5, 3, 11, "cross")]:
        X = rng.standard_normal((N1, L1, d)) * 0.5
        Y = rng.standard_normal((N2, L2, d)) * 0.5 if kind == "cross" else None
        G = rng.standard_normal((M + 1, N1) if kind == "diag" else (M + 1, N1, N2 if kind == "cross" else N1))
        kt = _t_kern(base, d, M, difference=difference)
        tX = torch.tensor(X, requires_grad=True)
        tY = None if Y is None else torch.tensor(Y, requires_grad=True)
        lev = kt.K_seq_diag_levels(tX) if kind == "diag" else kt.K_seq_levels(tX, tY)
        (lev * torch.tensor(G)).sum().backward()
        keep = []
        p = _params(base, d, M, difference, keep)
        gX, gY, gb = np.empty_like(X), (None if Y is None else np.empty_like(Y)), np.zeros(2)
        if kind == "diag":
            ctx.call("gpsig_seq_diag_levels_grad", p, _vp(X), N1, L1, _vp(G), _vp(gX), gb.ctypes.data_as(_P))
        else:
            ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N1, N2 if Y is not None else N1, L1, L2 if Y is not None else L1,
                     _vp(G), _vp(gX), _vp(gY), gb.ctypes.data_as(_P))
        assert rel(gX, tX.grad) < tol, (kind, rel(gX, tX.grad))
        if Y is not None:
            assert rel(gY, tY.grad) < tol
        if base in ("poly", "mix"):
            assert abs(gb[0] - kt.p0.grad.item()) < 1e-9 * max(1.0, abs(gb[0]))


@pytest.mark.parametrize("M,d", [(2, 1), (2, 5), (2, 32), (3, 2), (3, 6), (3, 16), (4, 3), (4, 4), (4, 8), (5, 2), (5, 4), (5, 8), (6, 3), (7, 2)])
@pytest.mark.parametrize("difference", [True, False])
@pytest.mark.parametrize("base", ["linear", "cosine"])
def test_linear_level_gradients_through_the_feature_contraction(M, d, difference, base):
    """Round 4: gradients of SignatureLinear's sequence levels through the explicit level features (csrc/sig_feat_grad_api.hip:
    features, one rocBLAS dgemm per level and side, the reverse sweep of sig_feat_grad_kernel.hpp).  Forced on
    (option sig_features_grad = 1) for cross / symmetric / diagonal calls over the shapes the feature kernels are built for --
    one- and several-wavefront workgroups, columns that do and do not divide the thread count, rows over several passes of the
    reduction tile (d = 16, 32) -- against autograd of the differentiable oracle and against the pair kernels' reverse pass.
    SignatureCosine takes the route as the linear kernel of the unit vectors x / |x|, the gradient taken on through that normalisation."""
    if base == "cosine" and (d == 1 or (M, d) in ((2, 32), (3, 16), (5, 8), (7, 2), (6, 3))):
        pytest.skip("cosine: one column has no gradient (kappa = +-1); a sample of the shapes is enough")
    rng = np.random.default_rng(100 * M + d)
    ctx = _host_ctx()
    shapes = [(7, 5, 9, 6, "cross"), (9, 9, 7, 7, "sym"), (11, 11, 5, 5, "diag"), (3, 4, 2, 12, "cross")]
    if d ** M <= 4096:
        shapes.append((70, 33, 17, 12, "cross"))
    try:
        for (N1, N2, L1, L2, kind) in shapes:
            if not difference and kind == "cross" and L1 == 2:
                L1 = 1                                               # a single observation: one increment-free step
            X = rng.standard_normal((N1, L1, d)) * 0.6
            Y = rng.standard_normal((N2, L2, d)) * 0.6 if kind == "cross" else None
            G = rng.standard_normal((M + 1, N1) if kind == "diag" else (M + 1, N1, N2 if kind == "cross" else N1))
            kt = _t_kern(base, d, M, difference=difference)
            tX = torch.tensor(X, requires_grad=True)
            tY = None if Y is None else torch.tensor(Y, requires_grad=True)
            lev = kt.K_seq_diag_levels(tX) if kind == "diag" else kt.K_seq_levels(tX, tY)
            (lev * torch.tensor(G)).sum().backward()
            keep = []
            p = _params(base, d, M, difference, keep)
            got = {}
            for route in (1, 0):
                ctx.set_option("sig_features_grad", route)
                gX, gY, gb = np.full_like(X, np.nan), (None if Y is None else np.full_like(Y, np.nan)), np.zeros(2)
                if kind == "diag":
                    ctx.call("gpsig_seq_diag_levels_grad", p, _vp(X), N1, L1, _vp(G), _vp(gX), gb.ctypes.data_as(_P))
                else:
                    ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N1, N2 if Y is not None else N1, L1, L2 if Y is not None else L1,
                             _vp(G), _vp(gX), _vp(gY), gb.ctypes.data_as(_P))
                got[route] = (gX, gY)
            for route in (1, 0):
                gX, gY = got[route]
                assert rel(gX, tX.grad) < 1e-9, (kind, route, N1, L1, rel(gX, tX.grad))
                if Y is not None:
                    assert rel(gY, tY.grad) < 1e-9, (kind, route, rel(gY, tY.grad))
            assert rel(got[1][0], got[0][0]) < 1e-10
    finally:
        ctx.set_option("sig_features_grad", -1)


@pytest.mark.parametrize("M,d,order", [(2, 3, 2), (3, 2, 2), (3, 2, 3), (3, 6, 2), (4, 3, 2), (4, 3, 3), (4, 3, 4), (4, 8, 4), (5, 2, 3), (5, 4, 5), (5, 8, 2),
                                       (5, 8, 5), (6, 3, 6), (6, 2, 4), (3, 16, 3), (2, 32, 2)])
@pytest.mark.parametrize("base", ["linear", "cosine"])
def test_higher_order_linear_gradients_through_the_feature_contraction(M, d, order, base):
    """Round 4: the higher-order algorithm (signature_algs.py:37-74; order = num_levels is the signature kernel the reference's notebook
    checks against esig) of the linear / cosine kernel differentiates through the feature contraction too: the step is a multiplication
    by the truncated exponential of the increment, undone by its inverse series, its adjoint taken through `order` Horner sub-steps
    (sig_feat_reverse_ho_kernel).  Cross / symmetric / diagonal calls against autograd of the differentiable oracle, and against the
    scratch-based higher-order kernels (which took 13 s for one 2,048 x 2,048 Gram at order 3: tools/bench_train_paths.py)."""
    if base == "cosine" and (M, d, order) not in ((3, 2, 2), (4, 3, 4), (5, 4, 5), (3, 6, 2)):
        pytest.skip("cosine: a sample of the shapes")
    rng = np.random.default_rng(1000 * order + 10 * M + d)
    ctx = _host_ctx()
    try:
        for (N1, N2, L1, L2, kind) in [(6, 5, 8, 5, "cross"), (7, 7, 6, 6, "sym"), (9, 9, 5, 5, "diag"), (3, 4, 2, 9, "cross")]:
            X = rng.standard_normal((N1, L1, d)) * 0.6
            Y = rng.standard_normal((N2, L2, d)) * 0.6 if kind == "cross" else None
            G = rng.standard_normal((M + 1, N1) if kind == "diag" else (M + 1, N1, N2 if kind == "cross" else N1))
            kt = _t_kern(base, d, M, difference=True, order=order)
            tX = torch.tensor(X, requires_grad=True)
            tY = None if Y is None else torch.tensor(Y, requires_grad=True)
            lev = kt.K_seq_diag_levels(tX) if kind == "diag" else kt.K_seq_levels(tX, tY)
            (lev * torch.tensor(G)).sum().backward()
            keep = []
            p = _params(base, d, M, True, keep, order=order)
            got = {}
            for route in ((1, 0) if d ** M <= 1024 else (1,)):
                ctx.set_option("sig_features_grad", route)
                gX, gY, gb = np.full_like(X, np.nan), (None if Y is None else np.full_like(Y, np.nan)), np.zeros(2)
                if kind == "diag":
                    ctx.call("gpsig_seq_diag_levels_grad", p, _vp(X), N1, L1, _vp(G), _vp(gX), gb.ctypes.data_as(_P))
                else:
                    ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N1, N2 if Y is not None else N1, L1, L2 if Y is not None else L1,
                             _vp(G), _vp(gX), _vp(gY), gb.ctypes.data_as(_P))
                got[route] = (gX, gY)
                assert rel(gX, tX.grad) < 1e-9, (kind, route, rel(gX, tX.grad))
                if Y is not None:
                    assert rel(gY, tY.grad) < 1e-9, (kind, route, rel(gY, tY.grad))
    finally:
        ctx.set_option("sig_features_grad", -1)


@pytest.mark.parametrize("base,d", [("rbf", 20), ("linear", 32), ("matern32", 40), ("rbf", 64)])
def test_higher_order_gradients_beyond_16_columns(base, d):
    """Regression (round 4): the higher-order gradient route pads 17 .. 64 columns to 32 / 64, and the contraction of Lam with the base
    kernel's derivatives used to fall into its 16-column instance for them -- the gradient's columns beyond 16 came out wrong, silently
    (found by the feature route's parity test at d = 32; the higher-order tests had stopped at 5 columns).  The scratch-based route
    (option sig_features_grad = 0) against the oracle's autograd, cross and symmetric."""
    rng = np.random.default_rng(d)
    ctx = _host_ctx()
    M, order = 2, 2
    try:
        ctx.set_option("sig_features_grad", 0)
        for (N1, N2, L1, L2, kind) in [(4, 3, 6, 4, "cross"), (5, 5, 5, 5, "sym")]:
            X = rng.standard_normal((N1, L1, d)) * 0.3
            Y = rng.standard_normal((N2, L2, d)) * 0.3 if kind == "cross" else None
            G = rng.standard_normal((M + 1, N1, N2 if kind == "cross" else N1))
            kt = _t_kern(base, d, M, difference=True, order=order)
            tX = torch.tensor(X, requires_grad=True)
            tY = None if Y is None else torch.tensor(Y, requires_grad=True)
            (kt.K_seq_levels(tX, tY) * torch.tensor(G)).sum().backward()
            keep = []
            p = _params(base, d, M, True, keep, order=order)
            gX, gY, gb = np.full_like(X, np.nan), (None if Y is None else np.full_like(Y, np.nan)), np.zeros(2)
            ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N1, N2 if Y is not None else N1, L1, L2 if Y is not None else L1,
                     _vp(G), _vp(gX), _vp(gY), gb.ctypes.data_as(_P))
            assert rel(gX, tX.grad) < 1e-9, (kind, rel(gX, tX.grad))
            if Y is not None:
                assert rel(gY, tY.grad) < 1e-9, (kind, rel(gY, tY.grad))
    finally:
        ctx.set_option("sig_features_grad", -1)


def test_feature_route_gradient_undoes_long_sweeps_accurately():
    """The reverse sweep stores nothing of the forward pass but its final features and UNDOES one step at a time: the early-time
    features come out as differences of the (much larger) late-time ones.  White-noise sequences of 150 and 300 points, whose level-5
    features reach 1e8 .. 1e11, against autograd of the differentiable oracle (which keeps every intermediate)."""
    rng = np.random.default_rng(77)
    ctx = _host_ctx()
    try:
        ctx.set_option("sig_features_grad", 1)
        for (N, L, d, M) in ((6, 150, 3, 5), (4, 300, 2, 6), (5, 128, 8, 4)):
            X = rng.standard_normal((N, L, d))
            G = rng.standard_normal((M + 1, N, N))
            kt = _t_kern("linear", d, M, difference=True)
            tX = torch.tensor(X, requires_grad=True)
            lev = kt.K_seq_levels(tX, None)
            (lev * torch.tensor(G)).sum().backward()
            keep = []
            p = _params("linear", d, M, True, keep)
            gX, gb = np.full_like(X, np.nan), np.zeros(2)
            ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), None, N, N, L, L, _vp(G), _vp(gX), None, gb.ctypes.data_as(_P))
            assert float(lev.detach().abs().max()) > 1e8
            assert rel(gX, tX.grad) < 1e-9, (N, L, d, M, rel(gX, tX.grad))
    finally:
        ctx.set_option("sig_features_grad", -1)


def test_linear_gram_gradient_through_the_feature_contraction_at_training_size():
    """The same route where it is the planner's own choice (option left at -1): a 512 x 512 symmetric Gram of sequences of 40 points in 4
    columns, 4 levels, against the pair kernels' reverse pass; and a module-level check -- normalised, weighted K(X) with lengthscales
    through SignatureKernelModule -- against the oracle's autograd."""
    from gpsig_amd import autodiff, kernels
    rng = np.random.default_rng(9)
    ctx = _host_ctx()
    N, L, d, M = 512, 40, 4, 4
    X = np.cumsum(0.3 * rng.standard_normal((N, L, d)), axis=1)
    G = rng.standard_normal((M + 1, N, N))
    keep = []
    p = _params("linear", d, M, True, keep)
    got = {}
    try:
        for route in (-1, 0):
            ctx.set_option("sig_features_grad", route)
            gX, gb = np.full_like(X, np.nan), np.zeros(2)
            ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), None, N, N, L, L, _vp(G), _vp(gX), None, gb.ctypes.data_as(_P))
            got[route] = gX
    finally:
        ctx.set_option("sig_features_grad", -1)
    assert np.isfinite(got[-1]).all()
    assert rel(got[-1], got[0]) < 1e-9, rel(got[-1], got[0])
    # through the module (scaling, normalisation, weights as torch ops around the level primitive), the route forced on the GPU context
    from gpsig_amd import _lib
    N, L, d, M = 96, 20, 3, 4
    mod, orc = _module_and_oracle("linear", d, M, L)
    Xs = np.cumsum(0.3 * rng.standard_normal((N, L, d)), axis=1).reshape(N, L * d)
    W = rng.standard_normal((N, N))
    dctx = _lib.context(0, torch.cuda.current_stream(torch.device("cuda:0")).cuda_stream)
    try:
        dctx.set_option("sig_features_grad", 1)
        xt = torch.tensor(Xs, device="cuda:0", requires_grad=True)
        (mod.K(xt) * torch.tensor(W, device="cuda:0")).sum().backward()
    finally:
        dctx.set_option("sig_features_grad", -1)
    xo = torch.tensor(Xs, requires_grad=True)
    (orc.K(xo) * torch.tensor(W)).sum().backward()
    assert rel(xt.grad, xo.grad) < 1e-8, rel(xt.grad, xo.grad)
    assert rel(mod.raw_lengthscales.grad, orc.lengthscales.grad * torch.sigmoid(mod.raw_lengthscales.detach().cpu())) < 1e-8


@pytest.mark.parametrize("base,normalization,difference,order,num_lags",
                         [("linear", True, True, 1, 0), ("linear", False, True, 1, 0), ("linear", True, False, 1, 1), ("cosine", True, True, 1, 0),
                          ("cosine", False, False, 1, 0), ("linear", True, True, 3, 0), ("linear", False, True, 2, 0), ("cosine", True, True, 4, 0)])
def test_level_sum_gradient_as_one_op(base, normalization, difference, order, num_lags):
    """gpsig_kernel_K_grad (autodiff._SeqGramSum): K(X) and K(X, X2) of the linear / cosine kernel with the level sum, the normalisation and
    the weights inside the op -- one product of the upstream with the features of all levels -- against the oracle's autograd of
    kernels.py:401-476 (sequences, variances, sigma, lengthscales, lags) and against the route through the level primitives."""
    from gpsig_amd import _lib
    d, M, L, N, N2 = 3, 4, 12, 40, 28
    mod, orc = _module_and_oracle(base, d, M, L, num_lags=num_lags, normalization=normalization, difference=difference, order=order)
    rng = np.random.default_rng(77)
    X = np.cumsum(0.3 * rng.standard_normal((N, L, d)), axis=1).reshape(N, L * d)
    X2 = np.cumsum(0.3 * rng.standard_normal((N2, L - 3, d)), axis=1).reshape(N2, (L - 3) * d)
    Wa, Wb = rng.standard_normal((N, N)), rng.standard_normal((N, N2))
    dev = torch.device("cuda:0")
    dctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
    names = ["raw_variances", "raw_sigma", "raw_lengthscales"] + (["raw_lags", "raw_gamma"] if num_lags else [])

    def run(sum_route):
        mod.sum_route = sum_route
        mod.zero_grad()
        xt = torch.tensor(X, device=dev, requires_grad=True)
        x2 = torch.tensor(X2, device=dev, requires_grad=True)
        Ka, Kb = mod.K(xt), mod.K(xt, x2)
        assert ("SeqGramSum" in type(Ka.grad_fn).__name__) == sum_route and ("SeqGramSum" in type(Kb.grad_fn).__name__) == sum_route
        ((Ka * torch.tensor(Wa, device=dev)).sum() + (Kb * torch.tensor(Wb, device=dev)).sum()).backward()
        return Ka.detach().cpu(), Kb.detach().cpu(), xt.grad.cpu(), x2.grad.cpu(), {n: getattr(mod, n).grad.cpu().clone() for n in names}

    try:
        dctx.set_option("sig_features_grad", 1)
        one = run(True)
        levels = run(False)
    finally:
        dctx.set_option("sig_features_grad", -1)
        mod.sum_route = True
    xo, x2o = torch.tensor(X, requires_grad=True), torch.tensor(X2, requires_grad=True)
    Koa, Kob = orc.K(xo), orc.K(xo, x2o)
    ((Koa * torch.tensor(Wa)).sum() + (Kob * torch.tensor(Wb)).sum()).backward()
    assert rel(one[0], Koa.detach()) < 1e-10 and rel(one[1], Kob.detach()) < 1e-10
    assert rel(one[2], xo.grad) < 1e-8, rel(one[2], xo.grad)
    assert rel(one[3], x2o.grad) < 1e-8, rel(one[3], x2o.grad)
    assert rel(one[2], levels[2]) < 1e-9 and rel(one[3], levels[3]) < 1e-9
    for n in names:
        assert rel(one[4][n], levels[4][n]) < 1e-8, (n, one[4][n], levels[4][n])
    # the oracle's leaves are the constrained values: d/d raw = d/d value * d value / d raw
    assert rel(one[4]["raw_variances"], orc.variances.grad * torch.sigmoid(mod.raw_variances.detach().cpu())) < 1e-8
    assert rel(one[4]["raw_sigma"], orc.sigma.grad * torch.sigmoid(mod.raw_sigma.detach().cpu())) < 1e-8


def test_level_sum_op_is_the_route_taken_and_can_be_declined():
    """The one-op route answers its probe before the forward pass commits to it: taken for the linear kernel at a training-size shape with
    the planner's own choice, declined for the RBF kernel and with option sig_features_grad = 0 (then K runs through the level primitives)."""
    from gpsig_amd import _lib, autodiff
    dev = torch.device("cuda:0")
    dctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(5)
    N, L, d, M = 256, 32, 4, 4
    X = torch.tensor(np.cumsum(0.3 * rng.standard_normal((N, L, d)), axis=1), device=dev)
    spec = autodiff._Spec("linear", M, True)
    assert autodiff._SeqGramSum.applies(X, None, spec, True)
    assert autodiff._SeqGramSum.applies(X, X[:100, :20].contiguous(), spec, False)
    assert not autodiff._SeqGramSum.applies(X, None, autodiff._Spec("rbf", M, True), True)
    try:
        dctx.set_option("sig_features_grad", 0)
        assert not autodiff._SeqGramSum.applies(X, None, spec, True)
    finally:
        dctx.set_option("sig_features_grad", -1)
    # at this size: value and gradient against the level route, planner's choice
    mod, _ = _module_and_oracle("linear", d, M, L)
    res = {}
    for sum_route in (True, False):
        mod.sum_route = sum_route
        mod.zero_grad()
        xt = X.reshape(N, L * d).clone().requires_grad_(True)
        K = mod.K(xt)
        (K * torch.cos(torch.arange(N * N, device=dev, dtype=torch.float64).reshape(N, N))).sum().backward()
        res[sum_route] = (K.detach().cpu(), xt.grad.cpu(), mod.raw_variances.grad.cpu().clone())
    mod.sum_route = True
    for a, b in zip(res[True], res[False]):
        assert rel(a, b) < 1e-9, rel(a, b)


def test_signature_features_are_the_truncated_signature_and_differentiate():
    """gpsig_seq_features: at order = num_levels with differences the level features are the signature of the piecewise-linear path (Chen's
    identity, the oracle's independent validator standing in for esig, notebook cells 6-7); at order 1 their inner products are the first-order
    levels of signature_algs.py:8-35; gpsig_seq_features_grad against central differences of a random functional."""
    from gpsig_amd import autodiff
    rng = np.random.default_rng(3)
    N, L, d, M = 6, 11, 3, 4
    X = np.cumsum(0.4 * rng.standard_normal((N, L, d)), axis=1)
    dev = torch.device("cuda:0")
    lev = autodiff.signature_features(torch.tensor(X, device=dev), M, order=M, difference=True)
    assert [tuple(a.shape) for a in lev] == [(N, d ** m) for m in range(1, M + 1)]
    got = torch.cat(lev, dim=1).cpu().numpy()
    want = np.stack([O.truncated_signature(x, M)[1:] for x in X])
    assert np.abs(got - want).max() < 1e-12 * max(1.0, np.abs(want).max())
    ko = O.SignatureKernelOracle(L * d, d, M, base="linear", order=1, normalization=False, difference=True, lengthscales=None)
    lev1 = autodiff.signature_features(torch.tensor(X, device=dev), M, order=1)
    Kl = ko.K(X.reshape(N, -1), return_levels=True)            # unit variances and sigma
    for m in range(1, M + 1):
        assert rel((lev1[m - 1] @ lev1[m - 1].T).cpu(), Kl[m]) < 1e-12
    for order, difference, unit in ((1, True, False), (3, True, False), (2, False, True)):
        W = [torch.tensor(rng.standard_normal((N, d ** m)), device=dev) for m in range(1, M + 1)]
        f = lambda xt: sum((a * w).sum() for a, w in zip(autodiff.signature_features(xt, M, order=order, difference=difference, unit_points=unit), W))
        xt = torch.tensor(X + (1.0 if unit else 0.0), device=dev, requires_grad=True)
        f(xt).backward()
        V = rng.standard_normal(X.shape)
        h = 1e-5
        with torch.no_grad():
            num = (f(xt + h * torch.tensor(V, device=dev)) - f(xt - h * torch.tensor(V, device=dev))).item() / (2 * h)
        ana = float((xt.grad.cpu().numpy() * V).sum())
        assert abs(num - ana) < 1e-7 * max(1.0, abs(ana)), (order, difference, unit, num, ana)
    with pytest.raises(NotImplementedError):
        autodiff.signature_features(torch.tensor(X, device=dev), 1)


@pytest.mark.parametrize("base,normalization,difference,order,increments,num_lags",
                         [("linear", True, True, 1, False, 0), ("linear", False, True, 1, True, 0), ("linear", True, False, 1, True, 1),
                          ("cosine", True, True, 1, True, 0), ("cosine", False, False, 1, False, 0), ("linear", True, True, 3, False, 0),
                          ("linear", False, True, 2, True, 0), ("cosine", True, True, 4, True, 0)])
def test_inducing_tensor_covariances_through_the_level_features(base, normalization, difference, order, increments, num_lags):
    """Linear / cosine kernel: Kzx = <rank-one tensor features, level features of the sequences> and the level diagonals as squared norms
    (autodiff feature_route) against the oracle's autograd of kernels.py:591-671 and against the tensor-vs-sequence recursions' route."""
    d, M, L, N, T = 3, 4, 10, 30, 7
    mod, orc = _module_and_oracle(base, d, M, L, num_lags=num_lags, normalization=normalization, difference=difference, order=order)
    rng = np.random.default_rng(12)
    de, lt = d * (num_lags + 1), M * (M + 1) // 2
    off = 1.0 if base == "cosine" else 0.0
    X = np.cumsum(0.3 * rng.standard_normal((N, L, d)), axis=1).reshape(N, L * d) + off
    Z = 0.5 * rng.standard_normal((lt, T, 2, de) if increments else (lt, T, de)) + off
    W = [rng.standard_normal(sh) for sh in ((T, T), (T, N), (N,), (T, N))]
    dev = torch.device("cuda:0")
    names = ["raw_variances", "raw_sigma", "raw_lengthscales"] + (["raw_lags", "raw_gamma"] if num_lags else [])

    def loss(m, Zt, Xt, cv):
        Kzz, Kzx, Kxx = m.K_tens_n_seq_covs(Zt, Xt, increments=increments)
        return (Kzz * cv(W[0])).sum() + (Kzx * cv(W[1])).sum() + (Kxx * cv(W[2])).sum() + (m.K_tens_vs_seq(Zt, Xt, increments=increments) * cv(W[3])).sum()

    from gpsig_amd import autodiff
    assert autodiff._SigFeatures.ld(mod._spec, de, L) > 0            # the route under test is the one the module takes

    def run(feature_route):
        mod.feature_route = "always" if feature_route else False       # (True would ask a work threshold these small shapes do not reach)
        mod.zero_grad()
        Zt, Xt = torch.tensor(Z, device=dev, requires_grad=True), torch.tensor(X, device=dev, requires_grad=True)
        l = loss(mod, Zt, Xt, lambda a: torch.tensor(a, device=dev))
        l.backward()
        return l.item(), Zt.grad.cpu(), Xt.grad.cpu(), {n: getattr(mod, n).grad.cpu().clone() for n in names}

    try:
        feat = run(True)
        rec = run(False)
    finally:
        mod.feature_route = True
    Zc, Xc = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True)
    lo = loss(orc, Zc, Xc, torch.tensor)
    lo.backward()
    assert abs(feat[0] - lo.item()) < 1e-10 * max(1.0, abs(lo.item())), (feat[0], lo.item())
    assert rel(feat[1], Zc.grad) < 1e-8 and rel(feat[2], Xc.grad) < 1e-8, (rel(feat[1], Zc.grad), rel(feat[2], Xc.grad))
    assert abs(feat[0] - rec[0]) < 1e-10 * max(1.0, abs(rec[0]))
    assert rel(feat[1], rec[1]) < 1e-8 and rel(feat[2], rec[2]) < 1e-8
    for n in names:
        assert rel(feat[3][n], rec[3][n]) < 1e-7, (n, feat[3][n], rec[3][n])
    # return_levels through the same features
    with torch.no_grad():
        mod.feature_route = "always"
        a = mod.K_tens_vs_seq(torch.tensor(Z, device=dev), torch.tensor(X, device=dev), return_levels=True, increments=increments)
        mod.feature_route = False
        b = mod.K_tens_vs_seq(torch.tensor(Z, device=dev), torch.tensor(X, device=dev), return_levels=True, increments=increments)
        mod.feature_route = True
    assert rel(a, b) < 1e-10
    # the default (True) takes the route by the recursion kernels' work: not at this size, yes with the threshold lowered
    mod.feature_route = True
    mod.K_tens_vs_seq(torch.tensor(Z, device=dev), torch.tensor(X, device=dev), increments=increments)
    # (order > 1: at every size; SignatureCosine at every size too since round 6 -- its recursion kernels are the run-time family: 2.1 against 51.6 ms at 4,096 sequences)
    assert (mod._phi(torch.tensor(X, device=dev).reshape(N, L, d), 1.0) is None) == (order == 1 and base == "linear")
    mod._phi_memo = None
    saved, mod.feature_route_min_work = mod.feature_route_min_work, 0.0
    try:
        assert mod._phi(mod.scale_sequences(mod._seq3(torch.tensor(X, device=dev), False)), 1.0) is not None
    finally:
        mod.feature_route_min_work = saved
        mod._phi_memo = None


@pytest.mark.parametrize("base,difference", [("rbf", True), ("poly", True), ("matern32", False)])
def test_point_kernel_gradients_in_blocks(base, difference):
    """seq_lam_undo_kernel + lam_contract_kernel over several blocks of pairs: a symmetric Gram visits the pairs beyond a
    block's own square once, with G[i][j] + G[j][i]; the base-kernel parameter gradient is summed over the blocks."""
    rng = np.random.default_rng(24)
    ctx = _host_ctx()
    for (M, N, L, d, kind) in [(4, 41, 20, 3, "sym"), (3, 500, 18, 2, "diag"), (4, 30, 20, 3, "cross")]:
        X = rng.standard_normal((N, L, d)) * 0.4
        Y = rng.standard_normal((N + 3, L - 2, d)) * 0.4 if kind == "cross" else None
        G = rng.standard_normal((M + 1, N) if kind == "diag" else (M + 1, N, N + 3 if kind == "cross" else N))
        kt = _t_kern(base, d, M, difference=difference)
        tX = torch.tensor(X, requires_grad=True)
        tY = None if Y is None else torch.tensor(Y, requires_grad=True)
        lev = kt.K_seq_diag_levels(tX) if kind == "diag" else kt.K_seq_levels(tX, tY)
        (lev * torch.tensor(G)).sum().backward()
        keep = []
        p = _params(base, d, M, difference, keep)
        for mb in (1, 4096):
            ctx.set_option("grad_scratch_mb", mb)
            gX, gY, gb = np.empty_like(X), (None if Y is None else np.empty_like(Y)), np.zeros(2)
            if kind == "diag":
                ctx.call("gpsig_seq_diag_levels_grad", p, _vp(X), N, L, _vp(G), _vp(gX), gb.ctypes.data_as(_P))
            else:
                ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N, N + 3 if Y is not None else N, L, L - 2 if Y is not None else L,
                         _vp(G), _vp(gX), _vp(gY), gb.ctypes.data_as(_P))
            ctx.set_option("grad_scratch_mb", 4096)
            assert rel(gX, tX.grad) < 1e-9, (kind, mb, rel(gX, tX.grad))
            if Y is not None:
                assert rel(gY, tY.grad) < 1e-9, (kind, mb)
            if base == "poly":
                assert abs(gb[0] - kt.p0.grad.item()) < 1e-9 * max(1.0, abs(gb[0])), (kind, mb)


def test_gradients_with_wide_state_space():
    """More than 32 feature columns (e.g. d = 11 with two lags): the one-pair-per-thread kernels at a padded width of 64."""
    rng = np.random.default_rng(26)
    ctx = _host_ctx()
    M, d = 3, 40
    for base, difference in (("rbf", True), ("linear", True), ("matern32", False)):
        X, Y = rng.standard_normal((5, 7, d)) * 0.2, rng.standard_normal((4, 6, d)) * 0.2
        G = rng.standard_normal((M + 1, 5, 4))
        kt = _t_kern(base, d, M, difference=difference)
        tX, tY = torch.tensor(X, requires_grad=True), torch.tensor(Y, requires_grad=True)
        (kt.K_seq_levels(tX, tY) * torch.tensor(G)).sum().backward()
        keep = []
        p = _params(base, d, M, difference, keep)
        gX, gY = np.empty_like(X), np.empty_like(Y)
        ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), 5, 4, 7, 6, _vp(G), _vp(gX), _vp(gY), None)
        assert rel(gX, tX.grad) < 1e-9 and rel(gY, tY.grad) < 1e-9, (base, rel(gX, tX.grad))
        # inducing tensors against sequences, and against each other
        T, lt = 6, M * (M + 1) // 2
        Z = rng.standard_normal((lt, T, 2, d)) * 0.2
        Gt = rng.standard_normal((M + 1, T, 5))
        tZ, tX = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True)
        (kt.K_tens_vs_seq_levels(tZ, tX, True) * torch.tensor(Gt)).sum().backward()
        gZ, gX = np.empty_like(Z), np.empty_like(X)
        ctx.call("gpsig_tens_vs_seq_levels_grad", p, _vp(Z), _vp(X), T, 5, 7, 1, _vp(Gt), _vp(gZ), _vp(gX), None)
        assert rel(gZ, tZ.grad) < 1e-9 and rel(gX, tX.grad) < 1e-9, (base, rel(gZ, tZ.grad))
        Gz = rng.standard_normal((M + 1, T, T))
        tZ = torch.tensor(Z, requires_grad=True)
        (kt.K_tens_levels(tZ, True) * torch.tensor(Gz)).sum().backward()
        gZ = np.empty_like(Z)
        ctx.call("gpsig_tens_gram_levels_grad", p, _vp(Z), T, 1, _vp(Gz), _vp(gZ), None)
        assert rel(gZ, tZ.grad) < 1e-9, (base, rel(gZ, tZ.grad))
    # beyond 64 columns: the wide route (round 6, csrc/wide_api.hip) for the distance kernels, an error for the others
    X = rng.standard_normal((2, 4, 65)) * 0.2
    G = rng.standard_normal((M + 1, 2, 2))
    gX = np.empty_like(X)
    keep = []
    p = _params("poly", 65, M, True, keep)
    with pytest.raises(NotImplementedError, match="at most 64 feature columns"):
        ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), None, 2, 2, 4, 4, _vp(G), _vp(gX), None, None)
    p = _params("rbf", 65, M, True, keep)
    ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), None, 2, 2, 4, 4, _vp(G), _vp(gX), None, None)
    tX = torch.tensor(X, requires_grad=True)
    (_t_kern("rbf", 65, M).K_seq_levels(tX, None) * torch.tensor(G)).sum().backward()
    assert rel(gX, tX.grad) < 1e-9


def test_seq_level_gradients_are_chunk_invariant():
    rng = np.random.default_rng(22)
    ctx = _host_ctx()
    M, N1, N2, L1, L2, d = 4, 100, 37, 12, 10, 4
    X, Y = rng.standard_normal((N1, L1, d)) * 0.4, rng.standard_normal((N2, L2, d)) * 0.4
    G = rng.standard_normal((M + 1, N1, N2))
    keep = []
    p = _params("rbf", d, M, True, keep)
    res = []
    for mb in (4096, 1):
        ctx.set_option("grad_scratch_mb", mb)
        gX, gY = np.empty_like(X), np.empty_like(Y)
        ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N1, N2, L1, L2, _vp(G), _vp(gX), _vp(gY), None)
        res.append((gX, gY))
    ctx.set_option("grad_scratch_mb", 4096)
    assert rel(res[1][0], res[0][0]) < 1e-12 and rel(res[1][1], res[0][1]) < 1e-12
    kt = _t_kern("rbf", d, M)
    tX, tY = torch.tensor(X, requires_grad=True), torch.tensor(Y, requires_grad=True)
    (kt.K_seq_levels(tX, tY) * torch.tensor(G)).sum().backward()
    assert rel(res[0][0], tX.grad) < 1e-9 and rel(res[0][1], tY.grad) < 1e-9


@pytest.mark.parametrize("base", BASES)
@pytest.mark.parametrize("difference", [True, False])
@pytest.mark.parametrize("increments", [False, True])
def test_tensor_level_gradients(base, difference, increments):
    rng = np.random.default_rng(23)
    ctx = _host_ctx()
    M, T, N, L, d = 4, 5, 70, 9, 3
    lt = M * (M + 1) // 2
    Z = rng.standard_normal((lt, T, 2, d) if increments else (lt, T, d)) * 0.5
    X = rng.standard_normal((N, L, d)) * 0.5
    G = rng.standard_normal((M + 1, T, N))
    kt = _t_kern(base, d, M, difference=difference)
    tZ, tX = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True)
    (kt.K_tens_vs_seq_levels(tZ, tX, increments) * torch.tensor(G)).sum().backward()
    keep = []
    p = _params(base, d, M, difference, keep)
    # tile kernel (default) / tensor lanes, one level per sweep (round 1) / one pair per thread with a stored lattice / one pair per thread, scratch-free
    for impl, tile in ((0, 1), (0, 0), (1, 1), (2, 1)):
        gZ, gX, gb = np.empty_like(Z), np.empty_like(X), np.zeros(2)
        ctx.set_option("grad_impl", impl)
        ctx.set_option("tvs_grad_tile", tile)
        ctx.set_option("grad_scratch_mb", 1)     # several launches
        ctx.call("gpsig_tens_vs_seq_levels_grad", p, _vp(Z), _vp(X), T, N, L, int(increments), _vp(G), _vp(gZ), _vp(gX), gb.ctypes.data_as(_P))
        ctx.set_option("grad_scratch_mb", 4096)
        ctx.set_option("grad_impl", 0)
        ctx.set_option("tvs_grad_tile", 1)
        assert rel(gZ, tZ.grad) < 1e-9 and rel(gX, tX.grad) < 1e-9, (impl, tile, rel(gZ, tZ.grad), rel(gX, tX.grad))
        if base in ("poly", "mix"):
            assert abs(gb[0] - kt.p0.grad.item()) < 1e-9 * max(1.0, abs(gb[0])), (impl, tile)
    G2 = rng.standard_normal((M + 1, T, T))
    kt = _t_kern(base, d, M)
    tZ = torch.tensor(Z, requires_grad=True)
    (kt.K_tens_levels(tZ, increments) * torch.tensor(G2)).sum().backward()
    gZ, gb = np.empty_like(Z), np.zeros(2)
    ctx.call("gpsig_tens_gram_levels_grad", p, _vp(Z), T, int(increments), _vp(G2), _vp(gZ), gb.ctypes.data_as(_P))
    assert rel(gZ, tZ.grad) < 1e-9
    if base in ("poly", "mix"):
        assert abs(gb[0] - kt.p0.grad.item()) < 1e-9 * max(1.0, abs(gb[0]))


@pytest.mark.parametrize("M,T,N,L,d", [(1, 3, 5, 4, 2), (2, 70, 9, 1, 4), (3, 33, 130, 2, 5), (4, 65, 37, 13, 6), (4, 130, 20, 5, 3), (5, 40, 17, 6, 7),
                                       (6, 9, 11, 7, 8), (4, 64, 64, 8, 1), (4, 130, 200, 50, 6)])
@pytest.mark.parametrize("base", ["linear", "rbf", "matern32", "poly", "matern12", "matern52"])
def test_tensor_vs_sequence_tile_gradient_kernel(M, T, N, L, d, base):
    """tvs_grad_tile_kernel (round 3: every level of a wave in ONE reverse sweep per sequence, d/dx summed in LDS, no atomics) against
    autograd of the oracle and against the round-1 kernels: 1 .. 6 levels (one to four waves per workgroup), widths 1 .. 8 (padded to
    4 / 6 / 8), ragged tensor and sequence counts across lane, run and flush boundaries, a single time step, differences on / off,
    increments on / off (lane pairs for the non-linear families, collapsed for the linear one), several sequence chunks."""
    if N * T > 10000 and base in ("poly", "matern12"):
        pytest.skip("the large case (two sequence chunks at a scratch budget of 1 MiB) runs for the compile-time families")
    if base in ("matern12", "matern52") and (M, d) in ((1, 2), (2, 4), (6, 8)):
        pytest.skip("a sample of the shapes is enough for the other two Matern families (one instruction stream, round 6)")
    rng = np.random.default_rng(1000 * M + T + d)
    ctx = _host_ctx()
    lt = M * (M + 1) // 2
    for difference in (True, False):
        for increments in (False, True):
            if L == 1 and difference:
                continue        # no increment to differentiate
            Z = rng.standard_normal((lt, T, 2, d) if increments else (lt, T, d)) * 0.5
            X = rng.standard_normal((N, L, d)) * 0.5
            G = rng.standard_normal((M + 1, T, N))
            kt = _t_kern(base, d, M, difference=difference)
            tZ, tX = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True)
            (kt.K_tens_vs_seq_levels(tZ, tX, increments) * torch.tensor(G)).sum().backward()
            keep = []
            p = _params(base, d, M, difference, keep)
            res = {}
            for tile, mb in ((1, 4096), (1, 1), (0, 4096)):
                if tile == 0 and M > 4:
                    continue            # the round-1 tensor-lane kernel is built for four levels
                gZ, gX, gb = np.empty_like(Z), np.empty_like(X), np.zeros(2)
                ctx.set_option("tvs_grad_tile", tile)
                ctx.set_option("grad_scratch_mb", mb)
                try:
                    ctx.call("gpsig_tens_vs_seq_levels_grad", p, _vp(Z), _vp(X), T, N, L, int(increments), _vp(G), _vp(gZ), _vp(gX), gb.ctypes.data_as(_P))
                finally:
                    ctx.set_option("tvs_grad_tile", 1)
                    ctx.set_option("grad_scratch_mb", 4096)
                # (Matern-1/2 in ONE column: d kappa / dx = -+ exp(-r), the sign taken as (x - z) / sqrt(x^2 + z^2 - 2 x z) -- for the closest of
                # 64 x 512 points the squared distance cancels to 1e-9 of its terms: oracle and kernel round it differently, 3e-8 apart)
                tol = 1e-6 if (base == "matern12" and d == 1) else 1e-9
                assert rel(gZ, tZ.grad) < tol and rel(gX, tX.grad) < tol, (tile, mb, difference, increments, rel(gZ, tZ.grad), rel(gX, tX.grad))
                if base == "poly":
                    assert abs(gb[0] - kt.p0.grad.item()) < 1e-9 * max(1.0, abs(gb[0])), (tile, mb, difference, increments)
                res[(tile, mb)] = (gZ, gX)


@pytest.mark.parametrize("base,M,T,N,L,d,order", [("rbf", 4, 70, 45, 9, 6, 1), ("linear", 3, 33, 20, 5, 4, 1), ("poly", 4, 9, 70, 6, 3, 1),
                                                   ("matern32", 2, 40, 12, 4, 2, 1), ("rbf", 4, 12, 9, 7, 12, 1), ("rbf", 3, 10, 8, 6, 3, 2),
                                                   ("linear", 5, 66, 10, 8, 7, 1)])
def test_weighted_tensor_vs_sequence_sum(base, M, T, N, L, d, order):
    """gpsig_tens_vs_seq_weighted / _grad (round 3): sum_m fac[n][m] level_m[t][n] with the level sum inside the kernel, and its
    gradients with respect to Z, X and the factors, against autograd of the oracle's level array -- through the tile kernels, through
    the level primitives (tile kernels off; 12 columns; order 2), host and device pointers."""
    from gpsig_amd import _lib
    rng = np.random.default_rng(7 * M + T)
    lt = M * (M + 1) // 2
    for increments in (False, True):
        Z = rng.standard_normal((lt, T, 2, d) if increments else (lt, T, d)) * 0.5
        X = rng.standard_normal((N, L, d)) * 0.5
        F = rng.uniform(0.5, 1.5, (N, M + 1))
        G = rng.standard_normal((T, N))
        kt = _t_kern(base, d, M, order=order)
        tZ, tX, tF = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True), torch.tensor(F, requires_grad=True)
        want = (kt.K_tens_vs_seq_levels(tZ, tX, increments) * tF.t()[:, None, :]).sum(0)
        (want * torch.tensor(G)).sum().backward()
        keep = []
        p = _params(base, d, M, True, keep, order=order)
        ctx = _host_ctx()
        for tile in (1, 0):
            ctx.set_option("tvs_grad_tile", tile)
            ctx.set_option("tvs_tile", -1 if tile else 0)
            try:
                out = np.empty((T, N))
                ctx.call("gpsig_tens_vs_seq_weighted", p, _vp(Z), _vp(X), T, N, L, int(increments), _vp(F), _vp(out), None, None)
                gZ, gX, gF, gb = np.empty_like(Z), np.empty_like(X), np.empty_like(F), np.zeros(2)
                ctx.call("gpsig_tens_vs_seq_weighted_grad", p, _vp(Z), _vp(X), T, N, L, int(increments), _vp(F), _vp(G), None, _vp(gZ), _vp(gX),
                         _vp(gF), gb.ctypes.data_as(_P))
            finally:
                ctx.set_option("tvs_grad_tile", 1)
                ctx.set_option("tvs_tile", -1)
            assert rel(out, want) < 1e-10, (tile, increments, rel(out, want))
            assert rel(gZ, tZ.grad) < 1e-9 and rel(gX, tX.grad) < 1e-9 and rel(gF, tF.grad) < 1e-9, (
                tile, increments, rel(gZ, tZ.grad), rel(gX, tX.grad), rel(gF, tF.grad))
            if base == "poly":
                assert abs(gb[0] - kt.p0.grad.item()) < 1e-9 * max(1.0, abs(gb[0])), (tile, increments)
        # device pointers (what gpsig_amd.autodiff passes)
        dev = torch.device("cuda:0")
        dctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
        dctx.set_pointer_mode(_lib.PTR_DEVICE)
        dZ, dX, dF, dG = (torch.tensor(a, device=dev) for a in (Z, X, F, G))
        dgZ, dgX, dgF, dgb = torch.empty_like(dZ), torch.empty_like(dX), torch.empty_like(dF), torch.zeros(2, dtype=torch.float64, device=dev)
        ptr = lambda t_: C.c_void_p(t_.data_ptr())
        # ... with the chain totals handed from the forward call to the reverse pass where the tile kernel leaves them
        aux = torch.empty(int(_lib.load().gpsig_tens_vs_seq_aux_elems(C.byref(p), T, N)), dtype=torch.float64, device=dev)
        dout, wrote = torch.empty((T, N), dtype=torch.float64, device=dev), C.c_int32(0)
        dctx.call("gpsig_tens_vs_seq_weighted", p, ptr(dZ), ptr(dX), T, N, L, int(increments), ptr(dF), ptr(dout), ptr(aux), C.byref(wrote))
        assert rel(dout, want) < 1e-10
        # (the tile kernel leaves them -- at most 8 columns, 32 tensors or more -- and, round 6, the wide route beyond 8 columns for the distance kernels)
        wide = order == 1 and d > 8 and base in ("rbf", "matern12", "matern32", "matern52")
        assert wrote.value == (1 if ((order == 1 and d <= 8 and T >= 32) or wide) else 0), (wrote.value, T, d, order)
        for use_aux in (False, True):
            if use_aux and not wrote.value:
                continue
            dctx.call("gpsig_tens_vs_seq_weighted_grad", p, ptr(dZ), ptr(dX), T, N, L, int(increments), ptr(dF), ptr(dG), ptr(aux) if use_aux else None,
                      ptr(dgZ), ptr(dgX), ptr(dgF), C.cast(dgb.data_ptr(), _P))
            torch.cuda.synchronize()
            assert rel(dgZ, tZ.grad) < 1e-9 and rel(dgX, tX.grad) < 1e-9 and rel(dgF, tF.grad) < 1e-9, (use_aux, rel(dgZ, tZ.grad), rel(dgX, tX.grad))
        kt.p0.grad = None


@pytest.mark.parametrize("base", ["linear", "rbf", "poly", "matern32"])
@pytest.mark.parametrize("order,difference", [(2, True), (3, True), (4, True), (2, False)])
def test_higher_order_level_gradients(base, order, difference):
    """Reverse mode of the higher-order algorithms (signature_algs.py:37-74 as lattice operations over blocks of pairs,
    :129-160 one pair per thread) against autograd of their torch restatement: cross / symmetric / diagonal sequence levels in
    several scratch blocks, tensor-vs-sequence levels with and without increments."""
    rng = np.random.default_rng(51)
    ctx = _host_ctx()
    M = 4
    for (N1, N2, L1, L2, d, kind) in [(9, 4, 7, 6, 3, "cross"), (6, 6, 5, 5, 2, "sym"), (11, 11, 6, 6, 5, "diag")]:
        X = rng.standard_normal((N1, L1, d)) * 0.5
        Y = rng.standard_normal((N2, L2, d)) * 0.5 if kind == "cross" else None
        G = rng.standard_normal((M + 1, N1) if kind == "diag" else (M + 1, N1, N2 if kind == "cross" else N1))
        kt = _t_kern(base, d, M, difference=difference, order=order)
        tX = torch.tensor(X, requires_grad=True)
        tY = None if Y is None else torch.tensor(Y, requires_grad=True)
        lev = kt.K_seq_diag_levels(tX) if kind == "diag" else kt.K_seq_levels(tX, tY)
        (lev * torch.tensor(G)).sum().backward()
        keep = []
        p = _params(base, d, M, difference, keep, order=order)
        for mb in (4096, 1):                       # one block of pairs / several
            gX, gY, gb = np.empty_like(X), (None if Y is None else np.empty_like(Y)), np.zeros(2)
            ctx.set_option("grad_scratch_mb", mb)
            if kind == "diag":
                ctx.call("gpsig_seq_diag_levels_grad", p, _vp(X), N1, L1, _vp(G), _vp(gX), gb.ctypes.data_as(_P))
            else:
                ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N1, N2 if Y is not None else N1, L1, L2 if Y is not None else L1,
                         _vp(G), _vp(gX), _vp(gY), gb.ctypes.data_as(_P))
            ctx.set_option("grad_scratch_mb", 4096)
            assert rel(gX, tX.grad) < 1e-9, (kind, mb, rel(gX, tX.grad))
            if Y is not None:
                assert rel(gY, tY.grad) < 1e-9
            if base == "poly":
                assert abs(gb[0] - kt.p0.grad.item()) < 1e-9 * max(1.0, abs(gb[0]))
    T, N, L, d = 5, 70, 8, 3
    lt = M * (M + 1) // 2
    for increments in (False, True):
        Z = rng.standard_normal((lt, T, 2, d) if increments else (lt, T, d)) * 0.5
        X = rng.standard_normal((N, L, d)) * 0.5
        G = rng.standard_normal((M + 1, T, N))
        kt = _t_kern(base, d, M, difference=difference, order=order)
        tZ, tX = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True)
        (kt.K_tens_vs_seq_levels(tZ, tX, increments) * torch.tensor(G)).sum().backward()
        keep = []
        p = _params(base, d, M, difference, keep, order=order)
        gZ, gX, gb = np.empty_like(Z), np.empty_like(X), np.zeros(2)
        ctx.call("gpsig_tens_vs_seq_levels_grad", p, _vp(Z), _vp(X), T, N, L, int(increments), _vp(G), _vp(gZ), _vp(gX), gb.ctypes.data_as(_P))
        assert rel(gZ, tZ.grad) < 1e-9 and rel(gX, tX.grad) < 1e-9, (increments, rel(gZ, tZ.grad), rel(gX, tX.grad))
        if base == "poly":
            assert abs(gb[0] - kt.p0.grad.item()) < 1e-9 * max(1.0, abs(gb[0]))


def _module_and_oracle(base, d, M, L, num_lags=0, normalization=True, difference=True, lengthscales=True, order=1):
    from gpsig_amd import kernels, autodiff
    cls = {"linear": kernels.SignatureLinear, "rbf": kernels.SignatureRBF, "poly": kernels.SignaturePoly, "mix": kernels.SignatureMix,
           "matern32": kernels.SignatureMatern32, "cosine": kernels.SignatureCosine}[base]
    rng = np.random.default_rng(31)
    kern = cls(L * d, d, M, normalization=normalization, difference=difference, num_lags=num_lags or None, order=order,
               lengthscales=(rng.uniform(0.8, 1.6, d) if lengthscales else None), variances=rng.uniform(0.5, 1.5, M + 1))
    kern.sigma = 1.3
    mod = autodiff.SignatureKernelModule(kern, device="cuda:0")
    leaf = lambda t: None if t is None else t.detach().cpu().clone().requires_grad_(True)
    orc = OT.SignatureKernelTorchOracle(d, M, base, variances=leaf(mod.variances), sigma=leaf(mod.sigma), lengthscales=leaf(mod.lengthscales),
                                        normalization=normalization, difference=difference, num_lags=num_lags,
                                        lags=leaf(mod.lags) if num_lags else None, gamma=leaf(mod.gamma) if num_lags else None,
                                        p0=leaf(mod.p0), p1=kern._current_base_params()[1], order=order)
    return mod, orc


def _constrained_grads(mod, loss):
    """d loss / d constrained hyper-parameters of the module"""
    names = ["variances", "sigma", "lengthscales", "p0"] + (["lags", "gamma"] if mod.kern.num_lags > 0 else [])
    vals = {n: getattr(mod, n) for n in names}
    vals = {n: v for n, v in vals.items() if v is not None}
    return vals


@pytest.mark.parametrize("base,num_lags,normalization,difference,order",
                         [("rbf", 0, True, True, 1), ("linear", 1, True, True, 1), ("poly", 0, False, True, 1), ("mix", 0, True, False, 1),
                          ("matern32", 2, True, True, 1), ("cosine", 0, True, True, 1), ("linear", 0, False, False, 1),
                          ("rbf", 1, True, True, 2), ("linear", 0, True, True, 3)])
def test_module_gradients_match_oracle_autograd(base, num_lags, normalization, difference, order):
    d, M, L, N, N2, T = 3, 3, 8, 7, 5, 4
    mod, orc = _module_and_oracle(base, d, M, L, num_lags, normalization, difference, lengthscales=(base != "cosine"), order=order)
    rng = np.random.default_rng(32)
    X, X2 = rng.standard_normal((N, L * d)) * 0.5, rng.standard_normal((N2, L * d)) * 0.5
    de = d * (num_lags + 1)
    lt = M * (M + 1) // 2
    for increments in (False, True):
        Z = rng.standard_normal((lt, T, 2, de) if increments else (lt, T, de)) * 0.5
        W1, W2, W3 = rng.standard_normal((T, T)), rng.standard_normal((T, N)), rng.standard_normal(N)
        Wk, Wc = rng.standard_normal((N, N)), rng.standard_normal((N, N2))
        dev = torch.device("cuda:0")
        cu = lambda a: torch.tensor(a, device=dev)
        Zg = torch.tensor(Z, device=dev, requires_grad=True)
        Xg = torch.tensor(X, device=dev, requires_grad=True)
        Kzz, Kzx, Kxx = mod.K_tens_n_seq_covs(Zg, Xg, increments=increments)
        loss = (Kzz * cu(W1)).sum() + (Kzx * cu(W2)).sum() + (Kxx * cu(W3)).sum() + (mod.K(Xg) * cu(Wk)).sum() + (mod.K(Xg, cu(X2)) * cu(Wc)).sum()
        mod.zero_grad()
        loss.backward()
        Zc = torch.tensor(Z, requires_grad=True)
        Xc = torch.tensor(X, requires_grad=True)
        for t in (orc.variances, orc.sigma, orc.lengthscales, orc.lags, orc.gamma, orc.p0):
            if t is not None and t.grad is not None:
                t.grad = None
        oKzz, oKzx, oKxx = orc.K_tens_n_seq_covs(Zc, Xc, increments=increments)
        oloss = (oKzz * torch.tensor(W1)).sum() + (oKzx * torch.tensor(W2)).sum() + (oKxx * torch.tensor(W3)).sum() + \
                (orc.K(Xc) * torch.tensor(Wk)).sum() + (orc.K(Xc, torch.tensor(X2)) * torch.tensor(Wc)).sum()
        oloss.backward()
        assert abs(loss.item() - oloss.item()) < 1e-9 * max(1.0, abs(oloss.item()))
        assert rel(Zg.grad, Zc.grad) < 1e-8
        assert rel(Xg.grad, Xc.grad) < 1e-8
        # hyper-parameters: compare in the unconstrained space through the same transforms
        pairs = [(mod.raw_variances, orc.variances, "pos"), (mod.raw_sigma, orc.sigma, "pos")]
        if mod.raw_lengthscales is not None:
            pairs.append((mod.raw_lengthscales, orc.lengthscales, "pos"))
        if num_lags:
            pairs += [(mod.raw_lags, orc.lags, "logistic"), (mod.raw_gamma, orc.gamma, "pos")]
        if mod.raw_p0 is not None:
            pairs.append((mod.raw_p0, orc.p0, "pos"))
        for raw, con, kind in pairs:
            r = raw.detach().cpu()
            jac = torch.sigmoid(r) if kind == "pos" else torch.sigmoid(r) * (1 - torch.sigmoid(r))
            want = con.grad * jac
            assert rel(raw.grad, want) < 1e-8, (kind, raw.grad, want)


@pytest.mark.parametrize("base", ["rbf", "linear"])
def test_float32_module_is_computed_in_float64_and_rounded(base):
    """A module after .float() (float32 parameters and data): the level primitives run on the float64 kernels, values and
    gradients come back as float32 and agree with the float64 module to float32 rounding (linear: the one-op level sum converts
    the same way; the level-feature route of Kzx is float64's)."""
    d, M, L, N, T = 3, 3, 8, 6, 4
    mod, _ = _module_and_oracle(base, d, M, L)
    rng = np.random.default_rng(33)
    X, Z = rng.standard_normal((N, L * d)) * 0.5, rng.standard_normal((M * (M + 1) // 2, T, 2, d)) * 0.5
    W = torch.tensor(rng.standard_normal((T, N)), device="cuda:0")
    res = {}
    for dt in (torch.float64, torch.float32):
        m = mod.double() if dt == torch.float64 else mod.float()
        Xg = torch.tensor(X, device="cuda:0", dtype=dt, requires_grad=True)
        Zg = torch.tensor(Z, device="cuda:0", dtype=dt, requires_grad=True)
        Kzz, Kzx, Kxx = m.K_tens_n_seq_covs(Zg, Xg, increments=True)
        assert Kzx.dtype == dt and Kzz.dtype == dt and Kxx.dtype == dt
        m.zero_grad()
        ((Kzx * W.to(dt)).sum() + Kzz.sum() + Kxx.sum() + m.K(Xg).sum()).backward()
        assert Xg.grad.dtype == dt and Zg.grad.dtype == dt and m.raw_lengthscales.grad.dtype == dt
        res[dt] = [t.detach().double().cpu() for t in (Kzx, Xg.grad, Zg.grad, m.raw_lengthscales.grad, m.raw_variances.grad)]
    mod.double()
    for a, b in zip(res[torch.float32], res[torch.float64]):
        assert rel(a, b) < 2e-5


def test_hyperparameters_can_be_trained():
    """A few Adam steps on a kernel-alignment loss decrease it, and write_back() moves the values into the inference path."""
    from gpsig_amd import kernels, autodiff
    rng = np.random.default_rng(33)
    N, L, d, M = 24, 10, 2, 3
    lab = np.repeat([0, 1], N // 2)
    X = np.cumsum(rng.standard_normal((N, L, d)) * 0.3, axis=1) + lab[:, None, None] * np.linspace(0, 1, L)[None, :, None]
    target = torch.tensor((lab[:, None] == lab[None, :]).astype(np.float64), device="cuda:0")
    kern = kernels.SignatureRBF(L * d, d, M, lengthscales=3.0)
    mod = autodiff.SignatureKernelModule(kern, device="cuda:0")
    Xg = torch.tensor(X.reshape(N, -1), device="cuda:0")
    opt = torch.optim.Adam(mod.parameters(), lr=0.05)
    losses = []
    for _ in range(15):
        opt.zero_grad()
        K = mod.K(Xg)
        loss = ((K / (M + 1) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    mod.write_back()
    K_inf = kern.K(Xg)
    assert rel(K_inf, mod.K(Xg)) < 1e-9


# ---- the trainable SVGP (gpsig/models.py:13-73) -------------------------------------------------------------------------
def _toy(N=40, L=12, d=2, seed=40):
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 2, N)
    t = np.linspace(0, 1, L)
    X = np.cumsum(rng.standard_normal((N, L, d)) * 0.15, axis=1)
    X[:, :, 0] += np.where(lab[:, None] == 1, np.sin(4 * t)[None], 0.0)
    return X.reshape(N, -1), lab.astype(np.float64)[:, None]


@pytest.mark.parametrize("lik,whiten,feat_kind", [("multiclass", True, "tensors"), ("bernoulli", False, "tensors"), ("gaussian", True, "sequences")])
def test_training_step_recorded_as_one_hip_graph(lik, whiten, feat_kind):
    """SVGPModule.fit(graph=True) -- forward, backward and the Adam update of one step replayed as one HIP graph -- follows the
    eager loop: the same minibatches give the same ELBO trace and the same parameters (the recorded kernels are the eager ones)."""
    from gpsig_amd import kernels, models, likelihoods as LK, inducing_variables as iv
    rng = np.random.default_rng(47)
    N, L, d, M, T, R = 36, 12, 2, 3, 7, 3
    X, Y = _toy(N, L, d)
    if lik == "multiclass":
        Y = rng.integers(0, R, (N, 1)).astype(np.float64)
    Xg, Yg = torch.tensor(X, device="cuda:0"), torch.tensor(Y, device="cuda:0")
    Z = rng.standard_normal((M * (M + 1) // 2, T, 2, d)) * 0.4
    Zs = np.cumsum(rng.standard_normal((T, 6, d)) * 0.2, axis=1)
    out = {}
    for graph in (False, True):
        kern = kernels.SignatureRBF(L * d, d, M, lengthscales=1.2, normalization=(feat_kind == "tensors"))
        feat = iv.InducingTensors(Z.copy(), M, increments=True) if feat_kind == "tensors" else iv.InducingSequences(Zs.copy(), M)
        likelihood = LK.MultiClass(R) if lik == "multiclass" else (LK.Bernoulli() if lik == "bernoulli" else LK.Gaussian(0.3, device="cuda:0"))
        model = models.SVGPModule(kern, feat, likelihood, num_latent=R if lik == "multiclass" else 1, whiten=whiten, num_data=N, device="cuda:0")
        trace = model.fit(Xg, Yg, iterations=12, lr=0.02, minibatch_size=16, seed=5, graph=graph)
        out[graph] = (np.asarray(trace), [p.detach().cpu().numpy().copy() for p in model.parameters()])
    assert len(out[True][0]) == 12 and np.isfinite(out[True][0]).all()
    # (Adam with capturable=True keeps its step count on the device and rounds its bias corrections differently from the
    # default implementation: 1e-8 relative per update, observed 2e-7 on the trace after 12 steps)
    assert np.abs(out[True][0] - out[False][0]).max() <= 1e-5 * np.abs(out[False][0]).max()
    for a, b in zip(out[True][1], out[False][1]):
        assert np.abs(a - b).max() <= 1e-4 * (np.abs(b).max() + 1e-12)


@pytest.mark.parametrize("lik,increments,learn_weights,whiten,q_diag",
                         [("bernoulli", True, False, True, False), ("gaussian", False, True, False, False), ("multiclass", True, False, True, True)])
def test_svgp_elbo_and_its_gradient(lik, increments, learn_weights, whiten, q_diag):
    from gpsig_amd import kernels, models, likelihoods as LK, inducing_variables as iv
    from oracle import sigkern_oracle as O, svgp_oracle as SO
    rng = np.random.default_rng(41)
    N, L, d, M, T = 30, 10, 2, 3, 6
    X, Y = _toy(N, L, d)
    R = 3 if lik == "multiclass" else 1
    if lik == "multiclass":
        Y = rng.integers(0, R, (N, 1)).astype(np.float64)
    lt = M * (M + 1) // 2
    Z = rng.standard_normal((lt, T, 2, d) if increments else (lt, T, d)) * 0.4
    kern = kernels.SignatureRBF(L * d, d, M, lengthscales=rng.uniform(0.8, 1.5, d), variances=rng.uniform(0.5, 1.5, M + 1))
    feat = iv.InducingTensors(Z, M, increments=increments, learn_weights=learn_weights)
    if learn_weights:
        feat.W = feat.W + 0.1 * rng.standard_normal(feat.W.shape)
    mk = lambda dev: LK.Bernoulli() if lik == "bernoulli" else (LK.Gaussian(0.4, device=dev) if lik == "gaussian" else LK.MultiClass(R))
    likelihood = mk("cuda:0")
    model = models.SVGPModule(kern, feat, likelihood, num_latent=R, q_diag=q_diag, whiten=whiten, num_data=3 * N, device="cuda:0")
    with torch.no_grad():
        model.q_mu.copy_(torch.tensor(rng.standard_normal((T, R)) * 0.3))
        if q_diag:
            model.q_sqrt.copy_(torch.tensor(rng.uniform(0.5, 1.2, (T, R))))
        else:
            model.q_sqrt.copy_(torch.tensor(np.tril(np.eye(T)[None] + 0.1 * rng.standard_normal((R, T, T)))))
    Xg, Yg = torch.tensor(X, device="cuda:0"), torch.tensor(Y, device="cuda:0")
    elbo = model.elbo(Xg, Yg)
    elbo.backward()

    # (1) value against the NumPy oracles
    ko = O.SignatureKernelOracle(L * d, d, M, base="rbf", lengthscales=kern.lengthscales, variances=kern.variances)
    Kzz, Kzx, Kxx = O.inducing_tensors_Kuu_Kuf_Kff(ko, Z, X, increments=increments, W=feat.W if learn_weights else None, jitter=1e-6)
    q_mu, q_sqrt = model.q_mu.detach().cpu().numpy(), model.q_sqrt.detach().cpu().numpy()
    fm, fv = SO.base_conditional(Kzx, Kzz, Kxx, q_mu, q_sqrt=q_sqrt, white=whiten)
    kl = SO.gauss_kl(q_mu, q_sqrt, None if whiten else Kzz)
    ve = {"bernoulli": lambda: SO.bernoulli_variational_expectations(fm, fv, Y), "gaussian": lambda: SO.gaussian_variational_expectations(fm, fv, Y, 0.4),
          "multiclass": lambda: SO.multiclass_variational_expectations(fm, fv, Y, R)}[lik]()
    want = ve.sum() * 3.0 - kl
    assert abs(elbo.item() - want) < 1e-8 * max(1.0, abs(want)), (elbo.item(), want)

    # (2) gradient against autograd of the differentiable oracle + the same dense algebra on the CPU
    leaf = lambda t: t.detach().cpu().clone().requires_grad_(True)
    km = model.kernel
    orc = OT.SignatureKernelTorchOracle(d, M, "rbf", variances=leaf(km.variances), sigma=leaf(km.sigma), lengthscales=leaf(km.lengthscales))
    Zc, qmu_c, qs_c = leaf(model.Z), leaf(model.q_mu), leaf(model.q_sqrt)
    Wc = leaf(model.W) if learn_weights else None
    if learn_weights:
        lev = orc.K_tens_levels(orc.scale_tensors(Zc, increments), increments) * orc._w()[:, None, None]
        Xs = orc.scale_sequences(torch.tensor(X).reshape(N, L, d))
        levx = orc.K_tens_vs_seq_levels(orc.scale_tensors(Zc, increments), Xs, increments)
        levx = levx / torch.sqrt(orc.K_seq_diag_levels(Xs) + 1e-6)[:, None, :] * orc._w()[:, None, None]
        oKzz = lev[0] + (Wc @ lev[1:] @ Wc.transpose(1, 2)).sum(0)
        oKzx = levx[0] + (Wc @ levx[1:]).sum(0)
        oKxx = orc._w().sum().expand(N)
    else:
        oKzz, oKzx, oKxx = orc.K_tens_n_seq_covs(Zc, torch.tensor(X), increments=increments)
    oKzz = oKzz + 1e-6 * torch.eye(T, dtype=torch.float64)
    oKxx = oKxx + 1e-6
    qs_eff = qs_c if q_diag else torch.tril(qs_c)
    ofm, ofv = models.base_conditional(oKzx, oKzz, oKxx, qmu_c, q_sqrt=qs_eff, white=whiten)
    okl = models.gauss_kl(qmu_c, qs_eff, None if whiten else oKzz)
    lik_c = mk("cpu")
    oelbo = lik_c.variational_expectations(ofm, ofv, torch.tensor(Y)).sum() * 3.0 - okl
    oelbo.backward()
    assert abs(oelbo.item() - elbo.item()) < 1e-8 * max(1.0, abs(oelbo.item()))
    assert rel(model.Z.grad, Zc.grad) < 1e-7
    assert rel(model.q_mu.grad, qmu_c.grad) < 1e-7
    assert rel(torch.tril(model.q_sqrt.grad) if not q_diag else model.q_sqrt.grad, torch.tril(qs_c.grad) if not q_diag else qs_c.grad) < 1e-7
    if learn_weights:
        assert rel(model.W.grad, Wc.grad) < 1e-7
    sig = lambda r: torch.sigmoid(r.detach().cpu())
    assert rel(km.raw_lengthscales.grad, orc.lengthscales.grad * sig(km.raw_lengthscales)) < 1e-7
    assert rel(km.raw_variances.grad, orc.variances.grad * sig(km.raw_variances)) < 1e-7
    assert rel(km.raw_sigma.grad, orc.sigma.grad * sig(km.raw_sigma)) < 1e-7


def test_svgp_fit_learns_a_toy_classification():
    from gpsig_amd import kernels, models, likelihoods as LK, inducing_variables as iv
    N, L, d, M, T = 60, 12, 2, 3, 8
    X, Y = _toy(N, L, d, seed=42)
    rng = np.random.default_rng(43)
    Z = 0.3 * rng.standard_normal((M * (M + 1) // 2, T, 2, d))
    kern = kernels.SignatureRBF(L * d, d, M, lengthscales=1.0)
    model = models.SVGPModule(kern, iv.InducingTensors(Z, M, increments=True), LK.Bernoulli(), device="cuda:0")
    Xg, Yg = torch.tensor(X, device="cuda:0"), torch.tensor(Y, device="cuda:0")
    trace = model.fit(Xg, Yg, iterations=60, lr=0.05)
    assert trace[-1] > trace[0] + 5.0, (trace[0], trace[-1])
    p, _ = model.predict_y(Xg)
    acc = float(((p > 0.5).double() == Yg).double().mean())
    assert acc > 0.8, acc
    # the trained values drive the fused inference path too
    model.kernel.write_back()
    sv = models.SVGP(kern, iv.InducingTensors(model.Z.detach().cpu().numpy(), M, increments=True), q_mu=model.q_mu.detach().cpu().numpy(),
                     q_sqrt=model.q_sqrt.detach().cpu().numpy())
    fm, fv = sv.predict_f(X)
    fm2, fv2 = model.predict_f(Xg)
    assert rel(fm, fm2) < 1e-8 and rel(fv, fv2) < 1e-8


@pytest.mark.parametrize("base,difference", [("linear", True), ("rbf", True), ("rbf", False), ("matern52", True), ("linear", False)])
def test_wave_and_storage_gradient_kernels_agree(base, difference):
    """The wavefront-parallel path (default) against the one-pair-per-thread storage path, over the kernel shapes the planner
    can pick: 16 or 64 lanes per pair, 2..8 columns per lane, padded feature counts 4 / 8 / 16, more than 5 levels."""
    rng = np.random.default_rng(50)
    ctx = _host_ctx()
    for (M, N1, N2, L1, L2, d, kind) in [(4, 9, 7, 20, 31, 3, "cross"), (5, 6, 6, 64, 64, 8, "sym"), (3, 70, 70, 50, 50, 6, "diag"), (7, 3, 4, 12, 100, 2, "cross"),
                                           (2, 2, 3, 40, 200, 12, "cross"), (3, 2, 2, 33, 400, 5, "cross"), (4, 5, 4, 1 + int(difference), 9, 3, "cross")]:
        X = rng.standard_normal((N1, L1, d)) * 0.3
        Y = rng.standard_normal((N2, L2, d)) * 0.3 if kind == "cross" else None
        G = rng.standard_normal((M + 1, N1) if kind == "diag" else (M + 1, N1, N2 if kind == "cross" else N1))
        keep = []
        p = _params(base, d, M, difference, keep)
        res = []
        for impl in (1, 0, 3, 4):   # one pair per thread (stored lattice) / planner's choice (point kernels: scratch-free sweeps with Lam out) / wavefront + stored lattice / wavefront scratch-free with the gradient formed in the sweep
            ctx.set_option("grad_impl", impl)
            gX, gY = np.empty_like(X), (None if Y is None else np.empty_like(Y))
            if kind == "diag":
                ctx.call("gpsig_seq_diag_levels_grad", p, _vp(X), N1, L1, _vp(G), _vp(gX), None)
            else:
                ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N1, N2 if Y is not None else N1, L1, L2 if Y is not None else L1,
                         _vp(G), _vp(gX), _vp(gY), None)
            res.append((gX, gY))
        ctx.set_option("grad_impl", 0)
        for k in (1, 2, 3):
            assert rel(res[k][0], res[0][0]) < 1e-9, (kind, L2, k, rel(res[k][0], res[0][0]))
            if Y is not None:
                assert rel(res[k][1], res[0][1]) < 1e-9, (kind, L2, k, rel(res[k][1], res[0][1]))


def test_inducing_sequences_model(K=None):
    """InducingSequences (inducing_variables.py:89-137): the differentiable K_seq_n_seq_covs equals the fused evaluation path
    (which is pinned to the oracle in test_gpu_parity.py), its gradient passes a finite-difference check, and the SVGP built on
    it trains."""
    from gpsig_amd import kernels, models, autodiff, likelihoods as LK, inducing_variables as iv
    rng = np.random.default_rng(60)
    N, L, d, M, T, Lz = 30, 10, 2, 3, 6, 4
    X, Y = _toy(N, L, d, seed=61)
    Zs = 0.4 * rng.standard_normal((T, Lz, d))
    kern = kernels.SignatureRBF(L * d, d, M, lengthscales=rng.uniform(0.8, 1.4, d), variances=rng.uniform(0.6, 1.4, M + 1))
    mod = autodiff.SignatureKernelModule(kern, device="cuda:0")
    dev = torch.device("cuda:0")
    Zg = torch.tensor(Zs, device=dev, requires_grad=True)
    Xg = torch.tensor(X, device=dev)
    for full in (False, True):
        got = mod.K_seq_n_seq_covs(Zg, Xg, full_X2_cov=full)
        want = kern.K_seq_n_seq_covs(Zs, X, full_X2_cov=full)
        for g, w in zip(got, want):
            assert rel(g, w) < 1e-9
    W1, W2 = torch.tensor(rng.standard_normal((T, T)), device=dev), torch.tensor(rng.standard_normal((T, N)), device=dev)

    def loss_of(Zt):
        a, b, c = mod.K_seq_n_seq_covs(Zt, Xg)
        return (a * W1).sum() + (b * W2).sum() + c.sum()
    loss_of(Zg).backward()
    h = 1e-6
    for idx in [(0, 0, 0), (2, 3, 1), (5, 1, 0)]:
        Zp, Zm = Zs.copy(), Zs.copy()
        Zp[idx] += h
        Zm[idx] -= h
        with torch.no_grad():
            fd = (loss_of(torch.tensor(Zp, device=dev)) - loss_of(torch.tensor(Zm, device=dev))).item() / (2 * h)
        assert abs(fd - Zg.grad[idx].item()) < 1e-5 * max(1.0, abs(fd)), (idx, fd, Zg.grad[idx].item())
    # Training: with normalisation the reference divides Kzx by the inducing-side diagonal twice (kernels.py:713 + :750,
    # reproduced), which makes Kzx inconsistent with Kzz and the predictive variance negative -- so train without it.
    kern2 = kernels.SignatureRBF(L * d, d, M, lengthscales=1.0, normalization=False, variances=0.3)
    model = models.SVGPModule(kern2, iv.InducingSequences(Zs, M, learn_weights=True), LK.Bernoulli(), device="cuda:0")
    trace = model.fit(Xg, torch.tensor(Y, device=dev), iterations=25, lr=0.02)
    assert np.all(np.isfinite(trace)) and trace[-1] > trace[0]


@pytest.mark.parametrize("base,normalization,order", [("linear", True, 1), ("linear", False, 1), ("cosine", True, 1), ("linear", True, 3)])
def test_inducing_sequence_covariances_through_the_level_features(base, normalization, order):
    """K_seq_n_seq_covs of the linear / cosine kernel (InducingSequences, kernels.py:674-761 with its double division): the cross block as ONE
    product of scaled level features, against the fused evaluation path and, with gradients, against the route through the level primitives."""
    from gpsig_amd import autodiff
    d, M, L, Lz, N, T = 3, 4, 12, 6, 40, 9
    mod, _ = _module_and_oracle(base, d, M, L, normalization=normalization, order=order)
    rng = np.random.default_rng(8)
    off = 1.0 if base == "cosine" else 0.0
    X = np.cumsum(0.3 * rng.standard_normal((N, L, d)), axis=1).reshape(N, L * d) + off
    Zs = 0.4 * rng.standard_normal((T, Lz, d)) + off
    dev = torch.device("cuda:0")
    assert autodiff._SigFeatures.ld(mod._spec, d, L) > 0 and autodiff._SigFeatures.ld(mod._spec, d, Lz) > 0
    names = ["raw_variances", "raw_sigma", "raw_lengthscales"]
    for full in (False, True):
        W = [torch.tensor(rng.standard_normal(sh), device=dev) for sh in ((T, T), (T, N), (N, N) if full else (N,))]
        res = {}
        for route in (True, False):
            mod.feature_route = "always" if route else False
            mod.zero_grad()
            Zg, Xg = torch.tensor(Zs, device=dev, requires_grad=True), torch.tensor(X, device=dev, requires_grad=True)
            out = mod.K_seq_n_seq_covs(Zg, Xg, full_X2_cov=full)
            sum((a * w).sum() for a, w in zip(out, W)).backward()
            res[route] = ([o.detach().cpu() for o in out], Zg.grad.cpu(), Xg.grad.cpu(), {n: getattr(mod, n).grad.cpu().clone() for n in names})
        mod.feature_route = True
        want = mod.kern.K_seq_n_seq_covs(Zs, X, full_X2_cov=full)
        for g, w_, r in zip(res[True][0], want, res[False][0]):
            assert rel(g, w_) < 1e-9 and rel(g, r) < 1e-10
        assert rel(res[True][1], res[False][1]) < 1e-8 and rel(res[True][2], res[False][2]) < 1e-8
        for n in names:
            assert rel(res[True][3][n], res[False][3][n]) < 1e-7, (full, n)


# ---- the matrix route (round 3): base-kernel tensors by torch GEMMs + autograd, recursions on their lattices in the library -----------
@pytest.mark.parametrize("order", [1, 2, 3])
def test_lattice_and_chain_primitives(order):
    """gpsig_lattice_levels / gpsig_chain_levels and their _grad against autograd of the oracle's signature_kern_* on the same
    increment lattices: several scratch chunks, single cells, empty lattices."""
    ctx = _host_ctx()
    rng = np.random.default_rng(60 + order)
    for M, P, R1, R2 in ((4, 37, 6, 5), (3, 5, 1, 7), (1, 9, 3, 3), (5, 70, 4, 4)):
        if order > M:
            continue
        dM = rng.standard_normal((P, R1, R2)) * 0.5
        G = rng.standard_normal((M + 1, P))
        t = torch.tensor(dM, requires_grad=True)
        M4 = t.reshape(P, R1, 1, R2)
        lev = (OT.signature_kern_first_order(M4, M, difference=False) if order == 1 else
               OT.signature_kern_higher_order(M4, M, order=order, difference=False))[:, :, 0]
        (lev * torch.tensor(G)).sum().backward()
        keep = []
        p = _params("linear", 1, M, True, keep, order=order)
        for mb in (4096, 1):
            ctx.set_option("grad_scratch_mb", mb)
            try:
                out, g = np.empty((M + 1, P)), np.empty_like(dM)
                ctx.call("gpsig_lattice_levels", p, _vp(dM), P, R1, R2, _vp(out))
                ctx.call("gpsig_lattice_levels_grad", p, _vp(dM), P, R1, R2, _vp(G), _vp(g))
            finally:
                ctx.set_option("grad_scratch_mb", 4096)
            assert rel(out, lev) < 1e-12 and rel(g, t.grad) < 1e-11, (M, P, order, mb, rel(out, lev), rel(g, t.grad))
        if order == 1:
            lt, R = M * (M + 1) // 2, R2
            m = rng.standard_normal((lt, R, P)) * 0.5
            tm = torch.tensor(m, requires_grad=True)
            lev = OT.signature_kern_tens_vs_seq_first_order(tm.permute(0, 2, 1)[:, :, None, :], M, difference=False)[:, :, 0]
            (lev * torch.tensor(G)).sum().backward()
            out, g = np.empty((M + 1, P)), np.empty_like(m)
            ctx.call("gpsig_chain_levels", p, _vp(m), P, R, _vp(out))
            ctx.call("gpsig_chain_levels_grad", p, _vp(m), P, R, _vp(G), _vp(g))
            assert rel(out, lev) < 1e-12 and rel(g, tm.grad) < 1e-11


def _compare_module_with_oracle(mod, orc, d_cols, M, L, extra_pairs, N=6, N2=4, T=5, tol=1e-8):
    rng = np.random.default_rng(33)
    d_in = mod.kern.num_features
    X, X2 = rng.standard_normal((N, L * d_in)) * 0.4, rng.standard_normal((N2, L * d_in)) * 0.4
    lt = M * (M + 1) // 2
    dev = torch.device("cuda:0")
    cu = lambda a: torch.tensor(a, device=dev)
    for increments in (False, True):
        Z = rng.standard_normal((lt, T, 2, d_cols) if increments else (lt, T, d_cols)) * 0.4
        W1, W2, W3 = rng.standard_normal((T, T)), rng.standard_normal((T, N)), rng.standard_normal(N)
        Wk, Wc = rng.standard_normal((N, N)), rng.standard_normal((N, N2))
        Zg, Xg = torch.tensor(Z, device=dev, requires_grad=True), torch.tensor(X, device=dev, requires_grad=True)
        Kzz, Kzx, Kxx = mod.K_tens_n_seq_covs(Zg, Xg, increments=increments)
        loss = (Kzz * cu(W1)).sum() + (Kzx * cu(W2)).sum() + (Kxx * cu(W3)).sum() + (mod.K(Xg) * cu(Wk)).sum() + (mod.K(Xg, cu(X2)) * cu(Wc)).sum() \
            + (mod.K_tens_vs_seq(Zg, Xg, increments=increments, return_levels=True)[1:] ** 2).sum()
        mod.zero_grad()
        loss.backward()
        Zc, Xc = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True)
        for _, con, _k in extra_pairs:
            con.grad = None
        for t in (orc.variances, orc.sigma, orc.lengthscales):
            if t is not None:
                t.grad = None
        oKzz, oKzx, oKxx = orc.K_tens_n_seq_covs(Zc, Xc, increments=increments)
        oloss = (oKzz * torch.tensor(W1)).sum() + (oKzx * torch.tensor(W2)).sum() + (oKxx * torch.tensor(W3)).sum() + \
            (orc.K(Xc) * torch.tensor(Wk)).sum() + (orc.K(Xc, torch.tensor(X2)) * torch.tensor(Wc)).sum() + \
            (orc.K_tens_vs_seq(Zc, Xc, increments=increments, return_levels=True)[1:] ** 2).sum()
        oloss.backward()
        assert abs(loss.item() - oloss.item()) < 1e-9 * max(1.0, abs(oloss.item())), (loss.item(), oloss.item())
        assert rel(Zg.grad, Zc.grad) < tol and rel(Xg.grad, Xc.grad) < tol, (rel(Zg.grad, Zc.grad), rel(Xg.grad, Xc.grad))
        pairs = [(mod.raw_variances, orc.variances, "pos"), (mod.raw_sigma, orc.sigma, "pos")] + list(extra_pairs)
        if mod.raw_lengthscales is not None:
            pairs.append((mod.raw_lengthscales, orc.lengthscales, "pos"))
        for raw, con, kind in pairs:
            r = raw.detach().cpu()
            jac = torch.sigmoid(r) if kind == "pos" else torch.sigmoid(r) * (1 - torch.sigmoid(r))
            assert rel(raw.grad, con.grad * jac) < tol, (kind, rel(raw.grad, con.grad * jac))


@pytest.mark.parametrize("family,normalization,difference,order", [("rbf", True, True, 1), ("exp", True, True, 1), ("mixed", False, True, 1),
                                                                   ("rbf", True, False, 1)])
def test_spectral_kernel_gradients(family, normalization, difference, order):
    """SignatureSpectral can be trained (alpha, omega, gamma: gpsig/kernels.py:912-914): every covariance of the module and its
    gradients with respect to Z, X, variances, sigma and the three spectral parameter arrays against autograd of the oracle --
    base-kernel tensors by torch ops, the lattice / chain recursions and their reverse passes by the library."""
    from gpsig_amd import kernels, autodiff
    d, M, L, Q = 3, 3, 7, 3
    rng = np.random.default_rng(35)
    kern = kernels.SignatureSpectral(L * d, d, M, family=family, Q=Q, normalization=normalization, difference=difference, order=order,
                                     variances=rng.uniform(0.5, 1.5, M + 1))
    kern.alpha, kern.omega, kern.gamma = np.exp(0.3 * rng.standard_normal(Q)), 0.3 * np.exp(0.3 * rng.standard_normal((Q, d))), np.exp(0.3 * rng.standard_normal((Q, d)))
    kern.sigma = 1.2
    mod = autodiff.SignatureKernelModule(kern, device="cuda:0")
    assert mod.matrix_route
    leaf = lambda t: t.detach().cpu().clone().requires_grad_(True)
    al, om, ga = leaf(autodiff.positive(mod.raw_alpha)), leaf(autodiff.positive(mod.raw_omega)), leaf(autodiff.positive(mod.raw_sgamma))
    orc = OT.SignatureKernelTorchOracle(d, M, "spectral", variances=leaf(mod.variances), sigma=leaf(mod.sigma), lengthscales=None,
                                        normalization=normalization, difference=difference, order=order, spectral=(al, om, ga, kern.family))
    _compare_module_with_oracle(mod, orc, d, M, L, [(mod.raw_alpha, al, "pos"), (mod.raw_omega, om, "pos"), (mod.raw_sgamma, ga, "pos")])
    # the trained values reach the evaluation path: the fused forward kernels on the written-back parameters give the module's numbers
    X = torch.tensor(rng.standard_normal((5, L * d)) * 0.4, device="cuda:0")
    with torch.no_grad():
        want = mod.K(X)
    got = mod.write_back().K(X)
    assert rel(got, want) < 1e-10


@pytest.mark.parametrize("base,d,num_lags,order", [("rbf", 70, 0, 1), ("linear", 40, 2, 1), ("matern32", 200, 0, 1), ("rbf", 100, 1, 1),
                                                   ("linear", 40, 2, 2), ("matern32", 70, 0, 3), ("rbf", 33, 1, 2)])
def test_gradients_beyond_64_columns(base, d, num_lags, order):
    """Training with state spaces the gradient kernels (64 columns after lags) do not reach: up to 256 columns and beyond through the
    matrix route, against autograd of the oracle (the reference's benchmarks run num_lags = 1 on state spaces of up to 963 features:
    benchmarks/run_gpsig_benchmarks.py:32).  order > 1 (round 6): the wide route's higher-order chains for the distance kernels, the matrix route's
    (array operations with torch's autograd) for the others; the sequence lattices through the matrix route's lattice op."""
    from gpsig_amd import kernels, autodiff
    M, L = 3, 6
    cls = {"linear": kernels.SignatureLinear, "rbf": kernels.SignatureRBF, "matern32": kernels.SignatureMatern32}[base]
    rng = np.random.default_rng(36)
    kern = cls(L * d, d, M, num_lags=num_lags or None, lengthscales=rng.uniform(0.8, 1.6, d) * np.sqrt(d), variances=rng.uniform(0.5, 1.5, M + 1), order=order)
    mod = autodiff.SignatureKernelModule(kern, device="cuda:0")
    # (round 6: the library's wide route takes the primitives it is built for -- the distance kernels --, the matrix route the rest)
    assert mod._mx("seq") if base == "linear" else not mod._mx("tvs")
    leaf = lambda t: None if t is None else t.detach().cpu().clone().requires_grad_(True)
    orc = OT.SignatureKernelTorchOracle(d, M, base, variances=leaf(mod.variances), sigma=leaf(mod.sigma), lengthscales=leaf(mod.lengthscales),
                                        num_lags=num_lags, lags=leaf(mod.lags) if num_lags else None, gamma=leaf(mod.gamma) if num_lags else None, order=order)
    extra = [(mod.raw_lags, orc.lags, "logistic"), (mod.raw_gamma, orc.gamma, "pos")] if num_lags else []
    _compare_module_with_oracle(mod, orc, d * (num_lags + 1), M, L, extra)


@pytest.mark.parametrize("base,sparsity,num_lags,normalization,difference",
                         [("rbf", "sqrt", 0, True, True), ("rbf", "lin", 1, True, True), ("matern32", "log", 0, False, True),
                          ("mix", "sqrt", 0, True, False), ("poly", "sqrt", 0, True, True)])
def test_low_rank_module_gradients(base, sparsity, num_lags, normalization, difference):
    """Gradients of LOW-RANK mode (kernels.py:239-311 and the low_rank branches of K / K_tens_vs_seq / K_tens_n_seq_covs, a training option
    of the reference's benchmark driver): the module's torch-op route -- landmarks gathered from the scaled inputs, whitening through an
    eigendecomposition, running sums, sparse projections -- against autograd of the checker's restatement given the same draw: values,
    d/dZ, d/dX (which includes the path through the landmarks), every hyper-parameter.  And the forward values against the HIP
    library's low-rank kernels fed the same landmarks / jitter / projections (kern.low_rank_state)."""
    from gpsig_amd import kernels, autodiff
    d, M, L, N, N2, T, c, r = 3, 3, 9, 8, 5, 4, 7, 6
    cls = {"rbf": kernels.SignatureRBF, "matern32": kernels.SignatureMatern32, "mix": kernels.SignatureMix, "poly": kernels.SignaturePoly}[base]
    rng = np.random.default_rng(77)
    kern = cls(L * d, d, M, normalization=normalization, difference=difference, num_lags=num_lags or None,
               lengthscales=rng.uniform(0.8, 1.6, d), variances=rng.uniform(0.5, 1.5, M + 1),
               low_rank=True, num_components=c, rank_bound=r, sparsity=sparsity)
    kern.sigma = 1.2
    kern.rng = np.random.default_rng(5)
    mod = autodiff.SignatureKernelModule(kern, device="cuda:0")
    leaf = lambda t: None if t is None else t.detach().cpu().clone().requires_grad_(True)
    orc = OT.LowRankTorchOracle(d, M, base, variances=leaf(mod.variances), sigma=leaf(mod.sigma), lengthscales=leaf(mod.lengthscales),
                                normalization=normalization, difference=difference, num_lags=num_lags,
                                lags=leaf(mod.lags) if num_lags else None, gamma=leaf(mod.gamma) if num_lags else None,
                                p0=leaf(mod.p0), p1=kern._current_base_params()[1])
    de = d * (num_lags + 1)
    lt = M * (M + 1) // 2
    X, X2 = rng.standard_normal((N, L * d)) * 0.5, rng.standard_normal((N2, L * d)) * 0.5
    dev = torch.device("cuda:0")
    cu = lambda a: torch.tensor(a, device=dev)
    for increments in (False, True):
        Z = rng.standard_normal((lt, T, 2, de) if increments else (lt, T, de)) * 0.5
        nz = lt * T * (2 if increments else 1)
        dr_c = mod.draw_low_rank(nz + N * L)            # Kzz / Kzx / Kxx: tensors' points, then the sequences'
        dr_k = mod.draw_low_rank(N * L)                 # K(X)
        dr_x = mod.draw_low_rank(N * L + N2 * L)        # K(X, X2)
        W1, W2, W3 = rng.standard_normal((T, T)), rng.standard_normal((T, N)), rng.standard_normal(N)
        Wk, Wc = rng.standard_normal((N, N)), rng.standard_normal((N, N2))
        Zg, Xg = torch.tensor(Z, device=dev, requires_grad=True), torch.tensor(X, device=dev, requires_grad=True)
        Kzz, Kzx, Kxx = mod.K_tens_n_seq_covs(Zg, Xg, increments=increments, lr=dr_c)
        Kk, Kc = mod.K(Xg, lr=dr_k), mod.K(Xg, cu(X2), lr=dr_x)
        loss = (Kzz * cu(W1)).sum() + (Kzx * cu(W2)).sum() + (Kxx * cu(W3)).sum() + (Kk * cu(Wk)).sum() + (Kc * cu(Wc)).sum()
        mod.zero_grad()
        loss.backward()
        Zc, Xc = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True)
        for t in (orc.variances, orc.sigma, orc.lengthscales, orc.lags, orc.gamma, orc.p0):
            if t is not None and t.grad is not None:
                t.grad = None
        oKzz, oKzx, oKxx = orc.set_draw(dr_c.idx, dr_c.jitter_diag, dr_c.sketches).K_tens_n_seq_covs(Zc, Xc, increments=increments)
        oKk = orc.set_draw(dr_k.idx, dr_k.jitter_diag, dr_k.sketches).K(Xc)
        oKc = orc.set_draw(dr_x.idx, dr_x.jitter_diag, dr_x.sketches).K(Xc, torch.tensor(X2))
        oloss = (oKzz * torch.tensor(W1)).sum() + (oKzx * torch.tensor(W2)).sum() + (oKxx * torch.tensor(W3)).sum() + \
                (oKk * torch.tensor(Wk)).sum() + (oKc * torch.tensor(Wc)).sum()
        oloss.backward()
        for a, b in ((Kzz, oKzz), (Kzx, oKzx), (Kxx, oKxx), (Kk, oKk), (Kc, oKc)):
            assert rel(a, b) < 1e-8, (increments, rel(a, b))
        # gradients pass through the eigendecomposition of a c x c Gram whose smallest eigenvalue gaps are ~1e-3: rocSOLVER against LAPACK
        assert rel(Zg.grad, Zc.grad) < 1e-6 and rel(Xg.grad, Xc.grad) < 1e-6, (rel(Zg.grad, Zc.grad), rel(Xg.grad, Xc.grad))
        pairs = [(mod.raw_variances, orc.variances, "pos"), (mod.raw_sigma, orc.sigma, "pos"), (mod.raw_lengthscales, orc.lengthscales, "pos")]
        if num_lags:
            pairs += [(mod.raw_lags, orc.lags, "logistic"), (mod.raw_gamma, orc.gamma, "pos")]
        if mod.raw_p0 is not None:
            pairs.append((mod.raw_p0, orc.p0, "pos"))
        for raw, con, kind in pairs:
            rr = raw.detach().cpu()
            jac = torch.sigmoid(rr) if kind == "pos" else torch.sigmoid(rr) * (1 - torch.sigmoid(rr))
            assert rel(raw.grad, con.grad * jac) < 1e-6, (kind, raw.grad, con.grad * jac)
        # the same numbers from the HIP library's low-rank kernels, given the landmarks this draw selects
        with torch.no_grad():
            pool = torch.cat([mod.scale_tensors(Zg).reshape(-1, de), mod.scale_sequences(mod._seq3(Xg)).reshape(-1, de)], dim=0)
            st = kern.low_rank_state(pool[torch.as_tensor(dr_c.idx, device=dev)].cpu().numpy(), dr_c.jitter_diag, dr_c.sketches)
        hzz, hzx = kern.K_tens(Z, increments=increments, lr_state=st), kern.K_tens_vs_seq(Z, X, increments=increments, lr_state=st)
        hxx = kern.Kdiag(X, lr_state=st)
        tol = 1e-7
        assert rel(hzz, Kzz) < tol and rel(hzx, Kzx) < tol and rel(hxx, Kxx) < tol, (rel(hzz, Kzz), rel(hzx, Kzx), rel(hxx, Kxx))
    # without lr=: a fresh draw per evaluation, as the reference's graph draws one; and an SVGP step in low-rank mode runs
    a, b = mod.K(Xg), mod.K(Xg)
    assert a.shape == (N, N) and not torch.equal(a, b) and bool(torch.isfinite(a).all())


@pytest.mark.parametrize("base,M,L,d,c,r,difference,sparsity",
                         [("rbf", 4, 50, 6, 50, 50, True, "sqrt"), ("rbf", 3, 9, 3, 7, 6, True, "sqrt"), ("linear", 3, 70, 4, 12, 20, True, "log"),
                          ("matern32", 5, 33, 2, 9, 5, False, "sqrt"), ("poly", 2, 17, 5, 20, 8, True, "lin"), ("mix", 1, 12, 3, 10, 4, True, "sqrt"),
                          ("rbf", 3, 130, 3, 16, 16, True, "sqrt"), ("cosine", 3, 20, 4, 6, 64, False, "sqrt"), ("matern52", 4, 1, 3, 5, 5, False, "sqrt"),
                          ("matern12", 3, 2, 3, 5, 7, True, "log")])
def test_low_rank_sequence_features_and_their_reverse_pass(base, M, L, d, c, r, difference, sparsity):
    """Round 4: gpsig_lr_seq_features_dev / gpsig_lr_seq_features_grad (csrc/lr_grad_api.hip, lr_grad_kernel.hpp) -- the low-rank feature
    map of a batch of sequences given landmarks and whitening on the device, and its reverse pass in one kernel (the forward sweep
    repeated into a per-workgroup scratch, the projections' adjoints as gathers over transposed copies of the projections, the base
    kernel's derivatives) -- against the torch-op route of round 3 (gather x gather x value + index_add, differentiated by autograd):
    the feature values, d/dX, d/d landmarks, d/d whitening and the base kernel's own parameter.  Shapes: BASELINE configs[2]'s at the
    reference's default ranks, several time chunks of 64, r > c and c > r, one and two observations, every base-kernel family."""
    from gpsig_amd import autodiff, kernels, low_rank as lrm
    rng = np.random.default_rng(1000 + M * 10 + d)
    dev = torch.device("cuda:0")
    cls = {"rbf": kernels.SignatureRBF, "linear": kernels.SignatureLinear, "matern32": kernels.SignatureMatern32, "poly": kernels.SignaturePoly,
           "mix": kernels.SignatureMix, "cosine": kernels.SignatureCosine, "matern52": kernels.SignatureMatern52, "matern12": kernels.SignatureMatern12}[base]
    kern = cls(L * d, d, M, difference=difference, lengthscales=None, low_rank=True, num_components=c, rank_bound=r, sparsity=sparsity)
    mod = autodiff.SignatureKernelModule(kern, device=dev)
    N = 37
    sk = lrm.draw_level_sketches(rng, M, c, r, sparsity)
    draw = autodiff.LowRankDraw(np.arange(c), 1e-6 * rng.random(c), sk)
    X0 = np.cumsum(0.4 * rng.standard_normal((N, L, d)), axis=1)
    S0 = 0.7 * rng.standard_normal((c, d))
    W0 = rng.standard_normal((c, c)) / np.sqrt(c)
    G0 = rng.standard_normal((N, 1 + c + (M - 1) * r))
    out = {}
    for route in ("hip", "torch"):
        X = torch.tensor(X0, device=dev, requires_grad=True)
        S = torch.tensor(S0, device=dev, requires_grad=True)
        Wh = torch.tensor(W0, device=dev, requires_grad=True)
        mod.zero_grad()
        scope = autodiff._LowRankScope.__new__(autodiff._LowRankScope)         # a scope around GIVEN landmarks / whitening
        scope.mod, scope.S, scope.Wh, scope._seq, scope._tens = mod, S, Wh, {}, {}
        _, _, scope.sk = draw.on(dev)
        scope.host_sketches = draw.sketches
        mod.lr_hip = route == "hip"
        try:
            Phi = torch.cat(scope.seq(X), dim=1)
        finally:
            mod.lr_hip = True
        (Phi * torch.tensor(G0, device=dev)).sum().backward()
        out[route] = (Phi.detach(), X.grad, S.grad, Wh.grad, None if mod.raw_p0 is None else mod.raw_p0.grad.clone())
    assert rel(out["hip"][0], out["torch"][0]) < 1e-11, rel(out["hip"][0], out["torch"][0])
    for k, name in ((1, "dX"), (2, "dS"), (3, "dWh")):
        assert rel(out["hip"][k], out["torch"][k]) < 1e-9, (name, rel(out["hip"][k], out["torch"][k]))
    if out["torch"][4] is not None:
        assert rel(out["hip"][4], out["torch"][4]) < 1e-9


def test_low_rank_svgp_trains():
    """The reference's benchmark driver can train in low-rank mode (benchmarks/models/train_gpsig.py:21, :58): an ELBO step through the
    low-rank covariances has finite gradients for every parameter and Adam decreases the loss."""
    from gpsig_amd import kernels, autodiff, models, inducing_variables, likelihoods
    rng = np.random.default_rng(8)
    N, L, d, M, T = 40, 12, 2, 3, 6
    lab = np.repeat([0, 1], N // 2)
    X = np.cumsum(rng.standard_normal((N, L, d)) * 0.3, axis=1) + lab[:, None, None] * np.linspace(0, 1, L)[None, :, None]
    kern = kernels.SignatureRBF(L * d, d, M, lengthscales=np.ones(d), low_rank=True, num_components=10, rank_bound=8)
    kern.rng = np.random.default_rng(3)
    Z = rng.standard_normal((M * (M + 1) // 2, T, 2, d)) * 0.5
    feat = inducing_variables.InducingTensors(Z, M, increments=True)
    m = models.SVGPModule(kern, feat, likelihoods.Bernoulli(), num_data=N, device="cuda:0")
    Xt = torch.tensor(X.reshape(N, -1), device="cuda:0")
    Yt = torch.tensor(lab[:, None].astype(np.float64), device="cuda:0")
    loss = -m.elbo(Xt, Yt)
    loss.backward()
    for n_, p_ in m.named_parameters():
        if p_.requires_grad:
            assert p_.grad is not None and bool(torch.isfinite(p_.grad).all()), n_
    trace = m.fit(Xt, Yt, iterations=30, lr=5e-2)
    assert np.isfinite(trace).all() and np.mean(trace[-5:]) > np.mean(trace[:5])


@pytest.mark.parametrize("M,N1,N2,L1,L2,d,kind", [
    (5, 6, 6, 64, 64, 8, "sym"),        # the bench's shape: every lane of a pair group busy, 63 lattice columns
    (2, 13, 13, 5, 5, 1, "sym"),        # one column of state space; 13 sequences: a ragged last quad of register-side sequences
    (6, 5, 9, 33, 64, 7, "cross"),      # six levels, seven features padded to eight
    (3, 1, 1, 2, 2, 4, "sym"),          # a single lattice cell
    (4, 3, 130, 64, 9, 5, "cross"),     # many register-side quads against three streamed sequences
    (5, 37, 37, 64, 64, 8, "sym"),      # runs of streamed sequences that start inside a quad's own square
    (3, 40, 2, 100, 50, 6, "cross"),    # more rows than a wavefront has lanes; half a quad
    (4, 9, 7, 20, 31, 3, "cross"),
    (4, 3, 5, 20, 100, 3, "cross"),     # the column side longer than 64 points, the row side not: the roles are exchanged
    (3, 6, 5, 150, 40, 2, "cross"),     # 150 rows: two workgroups per CU's worth of LDS
    (4, 5, 5, 130, 130, 3, "sym"),      # more than 64 points on the column side: 64 lanes per pair, one pair per wavefront
    (5, 3, 2, 100, 200, 8, "cross"),    # both sides beyond 64 points: the shorter one on the columns
    (2, 6, 6, 256, 256, 2, "sym"),      # the longest column side the kernel is built for
    (4, 5, 5, 20, 20, 12, "sym"),       # 9 .. 16 columns of state space: two lattice columns per lane
    (3, 3, 4, 40, 100, 16, "cross"),    # ... with 64 lanes per pair and the roles exchanged
    (5, 9, 9, 128, 128, 9, "sym"),      # ... at the longest column side of that form
    (5, 6, 6, 100, 100, 8, "sym"),      # 65 .. 128 points on the column side: 32 lanes per pair, two pairs per wavefront
    (3, 7, 5, 128, 70, 5, "cross"),
    (4, 5, 5, 50, 50, 12, "sym"),       # ... and 33 .. 64 points with two columns per lane
])
@pytest.mark.parametrize("base", ["rbf", "matern12", "matern32", "matern52"])
def test_stationary_kernels_reverse_pass_in_one_launch(base, M, N1, N2, L1, L2, d, kind):
    """grad_fused_kernel.hpp (round 5): SignatureRBF / SignatureMatern12 / 32 / 52 on points with differences, order 1 -- an evaluator and a
    sweeper wavefront per four pairs, Lam never leaving the chip -- is what the planner picks for these shapes.  Held to torch.autograd of the
    differentiable oracle at the contract's 1e-6 (observed 1e-13), and to the two older GPU routes (one pair per thread with the stored lattice;
    the sweeps with Lam through HBM + lam_contract_kernel) at 1e-9."""
    rng = np.random.default_rng(77)
    ctx = _host_ctx()
    X = np.cumsum(rng.standard_normal((N1, L1, d)) * 0.3, 1)
    Y = np.cumsum(rng.standard_normal((N2, L2, d)) * 0.3, 1) if kind == "cross" else None
    G = rng.standard_normal((M + 1, N1, N2 if kind == "cross" else N1))
    kt = _t_kern(base, d, M, difference=True)
    tX = torch.tensor(X, requires_grad=True)
    tY = None if Y is None else torch.tensor(Y, requires_grad=True)
    (kt.K_seq_levels(tX, tY) * torch.tensor(G)).sum().backward()
    keep = []
    p = _params(base, d, M, True, keep)
    res = []
    try:
        for impl in (0, 1, 4):
            ctx.set_option("grad_impl", impl)
            gX, gY = np.empty_like(X), (None if Y is None else np.empty_like(Y))
            ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N1, N2 if Y is not None else N1, L1, L2 if Y is not None else L1,
                     _vp(G), _vp(gX), _vp(gY), None)
            res.append((gX, gY))
    finally:
        ctx.set_option("grad_impl", 0)
    assert rel(res[0][0], tX.grad) < 1e-6, rel(res[0][0], tX.grad)
    if Y is not None:
        assert rel(res[0][1], tY.grad) < 1e-6, rel(res[0][1], tY.grad)
    for k in (1, 2):
        assert rel(res[0][0], res[k][0]) < 1e-9, (k, rel(res[0][0], res[k][0]))
        if Y is not None:
            assert rel(res[0][1], res[k][1]) < 1e-9, (k, rel(res[0][1], res[k][1]))


@pytest.mark.parametrize("M,N1,N2,L1,L2,d,kind", [(5, 6, 6, 64, 64, 8, "sym"), (3, 7, 5, 9, 33, 3, "cross"), (4, 5, 5, 100, 100, 2, "sym"), (2, 3, 9, 30, 200, 4, "cross"),
                                                  (6, 4, 4, 2, 2, 1, "sym")])
@pytest.mark.parametrize("base", ["rbf", "matern12", "matern32", "matern52"])
def test_stationary_kernels_reverse_pass_in_one_launch_without_differences(base, M, N1, N2, L1, L2, d, kind):
    """The same kernel with difference=False (the lattice is the kernel matrix of the points itself: no increments along either side, no
    neighbour values in the evaluation, the adjoint of the kernel values is Lam itself)."""
    rng = np.random.default_rng(79)
    ctx = _host_ctx()
    X = np.cumsum(rng.standard_normal((N1, L1, d)) * 0.3, 1)
    Y = np.cumsum(rng.standard_normal((N2, L2, d)) * 0.3, 1) if kind == "cross" else None
    G = rng.standard_normal((M + 1, N1, N2 if kind == "cross" else N1)) * 0.01       # (without differences the levels grow like L^m)
    kt = _t_kern(base, d, M, difference=False)
    tX = torch.tensor(X, requires_grad=True)
    tY = None if Y is None else torch.tensor(Y, requires_grad=True)
    (kt.K_seq_levels(tX, tY) * torch.tensor(G)).sum().backward()
    keep = []
    p = _params(base, d, M, False, keep)
    res = []
    try:
        for impl in (0, 1, 4):
            ctx.set_option("grad_impl", impl)
            gX, gY = np.empty_like(X), (None if Y is None else np.empty_like(Y))
            ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N1, N2 if Y is not None else N1, L1, L2 if Y is not None else L1,
                     _vp(G), _vp(gX), _vp(gY), None)
            res.append((gX, gY))
    finally:
        ctx.set_option("grad_impl", 0)
    assert rel(res[0][0], tX.grad) < 1e-6, rel(res[0][0], tX.grad)
    if Y is not None:
        assert rel(res[0][1], tY.grad) < 1e-6, rel(res[0][1], tY.grad)
    for k in (1, 2):
        assert rel(res[0][0], res[k][0]) < 1e-9, (k, rel(res[0][0], res[k][0]))
        if Y is not None:
            assert rel(res[0][1], res[k][1]) < 1e-9, (k, rel(res[0][1], res[k][1]))


def test_rbf_reverse_pass_in_one_launch_is_the_route_taken():
    """The fused kernel must be what runs at the bench's shape (a silent fall-back to the Lam-through-HBM sweeps is 2.7 times slower and would
    pass every parity test): 512 sequences of 64 x 8, forward + backward, timed against the older route on the same box."""
    import time
    rng = np.random.default_rng(78)
    ctx = _host_ctx()
    M, N, L, d = 5, 512, 64, 8
    X = np.cumsum(rng.standard_normal((N, L, d)) * 0.3, 1)
    G = rng.standard_normal((M + 1, N, N))
    keep = []
    p = _params("rbf", d, M, True, keep)
    gX = np.empty_like(X)
    t = {}
    try:
        for impl in (0, 4):
            ctx.set_option("grad_impl", impl)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), None, N, N, L, L, _vp(G), _vp(gX), None, None)
                best = min(best, time.perf_counter() - t0)
            t[impl] = best
    finally:
        ctx.set_option("grad_impl", 0)
    assert t[0] < 0.75 * t[4], t


@pytest.mark.parametrize("M,N1,N2,L1,L2,d,kind", [(5, 64, 64, 64, 64, 8, "sym"), (4, 130, 130, 33, 33, 5, "sym"), (5, 37, 41, 64, 64, 8, "cross"),
                                                  (4, 9, 7, 20, 31, 3, "cross"), (3, 10, 10, 12, 12, 4, "sym")])
@pytest.mark.parametrize("base", ["rbf", "matern12", "matern32", "matern52"])
def test_forward_pass_keeps_what_its_reverse_pass_needs(base, M, N1, N2, L1, L2, d, kind):
    """gpsig_seq_gram_levels_stash / _grad_stash through autodiff._SeqGramLevels (round 5): the evaluation kernel's stash instances write the
    forward recursion's row totals and final states, the backward call runs the fused reverse kernel's backward sweep only.  Where the library
    keeps nothing (shapes outside those instances: the last two cases) the route falls back by itself.  Held to torch.autograd of the oracle at
    1e-6 and to the recompute route (option grad_stash_mb = 0) at 1e-9; a stash overwritten by a later evaluation is noticed and not used."""
    from gpsig_amd import _lib
    from gpsig_amd.autodiff import _SeqGramLevels, _Spec
    dev = torch.device("cuda:0")
    dctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(81)
    X = np.cumsum(rng.standard_normal((N1, L1, d)) * 0.3, 1)
    Y = np.cumsum(rng.standard_normal((N2, L2, d)) * 0.3, 1) if kind == "cross" else None
    G = rng.standard_normal((M + 1, N1, N2 if kind == "cross" else N1))
    kt = _t_kern(base, d, M, difference=True)
    tX = torch.tensor(X, requires_grad=True)
    tY = None if Y is None else torch.tensor(Y, requires_grad=True)
    (kt.K_seq_levels(tX, tY) * torch.tensor(G)).sum().backward()
    spec = _Spec(base, M, True, 0.0, order=1)
    Gd = torch.tensor(G, device=dev)
    res, kept = [], []
    try:
        for mb, overwrite in ((4096, False), (0, False), (4096, True)):
            dctx.set_option("grad_stash_mb", mb)
            Xg = torch.tensor(X, device=dev, requires_grad=True)
            Yg = None if Y is None else torch.tensor(Y, device=dev, requires_grad=True)
            lev = _SeqGramLevels.apply(Xg, Yg, None, spec)
            kept.append(lev.grad_fn.stash is not None)
            if overwrite:                                   # another differentiated evaluation in between: this one's stash is gone
                X2 = torch.tensor(X[::-1].copy(), device=dev, requires_grad=True)
                _SeqGramLevels.apply(X2, None if Y is None else torch.tensor(Y, device=dev), None, spec)
            (lev * Gd).sum().backward()
            res.append((Xg.grad.cpu(), None if Yg is None else Yg.grad.cpu()))
    finally:
        dctx.set_option("grad_stash_mb", 4096)
    assert kept[1] is False
    if (M, L1, d) in ((5, 64, 8), (4, 33, 5)):
        assert kept[0] and kept[2]                          # the instances the fused reverse kernel continues from
    for gX, gY in res:
        assert rel(gX, tX.grad) < 1e-6
        assert rel(gX, res[1][0]) < 1e-9
        if Y is not None:
            assert rel(gY, tY.grad) < 1e-6 and rel(gY, res[1][1]) < 1e-9


@pytest.mark.parametrize("M,order,N1,N2,L1,L2,d,kind", [(4, 2, 9, 4, 7, 6, 3, "cross"), (4, 2, 40, 40, 50, 50, 6, "diag"), (5, 2, 6, 5, 33, 70, 4, "cross"),
                                                       (3, 3, 7, 7, 12, 12, 2, "sym"), (4, 3, 5, 6, 20, 31, 5, "cross"), (4, 4, 30, 30, 18, 18, 3, "diag"),
                                                       (5, 4, 4, 3, 9, 140, 2, "cross"), (2, 2, 6, 6, 8, 8, 20, "sym"), (5, 5, 3, 4, 11, 10, 3, "cross"),
                                                       (3, 2, 2, 2, 5, 300, 2, "cross"), (4, 2, 300, 300, 6, 6, 3, "diag"),
                                                       # 1,499 lattice rows: the row totals exceed the scratch-free sweeps' LDS, the planner takes the sweeps with HBM slots;
                                                       # 599 lattice columns: beyond every sweep instance, the lattice operations
                                                       (4, 2, 2, 3, 1500, 20, 2, "cross"), (3, 2, 2, 2, 6, 600, 2, "cross")])
@pytest.mark.parametrize("base", ["rbf", "matern32", "linear"])
def test_higher_order_reverse_pass_in_two_sweeps(M, order, N1, N2, L1, L2, d, kind, base):
    """Round 6: the reverse pass of the higher-order sequence recursion (signature_algs.py:37-74) as two skewed sweeps of a wavefront per pair
    (csrc/grad_wave_ho_kernel.hpp: the backward sweep undoes the forward one row by row and level by level within a cell, rebuilds the cell's grids,
    runs the adjoints down the levels; option grad_impl = 3: the prefixes a cell reads kept per cell in an HBM slot instead)
    against autograd of the oracle and against the lattice operations they replace (option grad_impl = 1): every lane shape (lattices of 5 .. 299
    columns), orders 2-4 (and order >= num_levels), 2-5 levels, several pair blocks and several rounds of the pair groups."""
    if base != "rbf" and ((M, order, kind) in ((5, 4, "cross"), (2, 2, "sym"), (3, 2, "cross"), (4, 2, "diag")) and N1 != 40 or L1 == 1500):
        pytest.skip("a sample of the shapes is enough for the other families")
    rng = np.random.default_rng(7 * M + order + L2)
    ctx = _host_ctx()
    for difference in ((True, False) if L2 < 100 else (True,)):
        X = rng.standard_normal((N1, L1, d)) * (0.5 if difference else 0.15)
        Y = rng.standard_normal((N2, L2, d)) * (0.5 if difference else 0.15) if kind == "cross" else None
        G = rng.standard_normal((M + 1, N1) if kind == "diag" else (M + 1, N1, N2 if kind == "cross" else N1))
        kt = _t_kern(base, d, M, difference=difference, order=order)
        tX = torch.tensor(X, requires_grad=True)
        tY = None if Y is None else torch.tensor(Y, requires_grad=True)
        lev = kt.K_seq_diag_levels(tX) if kind == "diag" else kt.K_seq_levels(tX, tY)
        (lev * torch.tensor(G)).sum().backward()
        keep = []
        p = _params(base, d, M, difference, keep, order=order)
        got = {}
        try:
            # scratch-free sweeps / sweeps through HBM slots / lattice operations; wide = 0: the point route's dM and contraction kernels around the
            # sweeps instead of the wide route's dgemms (which take RBF and the Matern families)
            for impl, mb, wide in ((0, 4096, -1), (0, 1, -1), (3, 4096, -1), (3, 1, 0), (0, 4096, 0), (1, 4096, -1)):
                ctx.set_option("grad_impl", impl)
                ctx.set_option("grad_scratch_mb", mb)
                ctx.set_option("wide", wide)
                ctx.set_option("wide_chunk_mb", 1 if mb == 1 else 0)
                gX, gY, gb = np.full_like(X, np.nan), (None if Y is None else np.full_like(Y, np.nan)), np.zeros(2)
                if kind == "diag":
                    ctx.call("gpsig_seq_diag_levels_grad", p, _vp(X), N1, L1, _vp(G), _vp(gX), gb.ctypes.data_as(_P))
                else:
                    ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N1, N2 if Y is not None else N1, L1, L2 if Y is not None else L1,
                             _vp(G), _vp(gX), _vp(gY), gb.ctypes.data_as(_P))
                got[(impl, mb, wide)] = (gX, gY)
        finally:
            ctx.set_option("grad_impl", 0)
            ctx.set_option("grad_scratch_mb", 4096)
            ctx.set_option("wide", -1)
            ctx.set_option("wide_chunk_mb", 0)
        for key, (gX, gY) in got.items():
            assert rel(gX, tX.grad) < 1e-9, (difference, key, rel(gX, tX.grad))
            if Y is not None:
                assert rel(gY, tY.grad) < 1e-9, (difference, key, rel(gY, tY.grad))
        assert rel(got[(0, 4096, -1)][0], got[(1, 4096, -1)][0]) < 1e-10
