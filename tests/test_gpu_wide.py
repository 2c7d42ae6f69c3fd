"""Round 6: the wide-state-space route (csrc/wide_api.hip, wide_kernels.hpp) through the C ABI against the oracles.

The reference's own benchmark settings (benchmarks/run_gpsig_benchmarks.py:32: num_lags=1 on time-augmented data) give state spaces of
2 (n_features + 1) columns: 10, 14, 26, 28, 46, 126, 1,928 for 9 of its 16 data sets.  Shapes below follow them (12 / 16 / 28 / 46 / 126 columns; the
route forced on at 3 and 8 as well); values against oracle/sigkern_oracle.py, gradients against autograd of oracle/sigkern_oracle_torch.py.
Tolerances: 1e-9 relative to the largest entry (float64; observed ~1e-13) -- inside north_star's 1e-6."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import sigkern_oracle as O
from oracle import sigkern_oracle_torch as OT

pytestmark = pytest.mark.gpu
_P = C.POINTER(C.c_double)
WIDE_BASES = ["rbf", "matern12", "matern32", "matern52"]


def rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def _host_ctx():
    from gpsig_amd import _lib
    ctx = _lib.context(0, 0)
    ctx.set_pointer_mode(_lib.PTR_HOST)
    return ctx


def _params(base, d, M, difference, keep):
    from gpsig_amd.autodiff import _Spec
    return _Spec(base, M, difference, 0.0, order=1).params(d, 0.0, keep)


def _vp(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _data(rng, M, T, N, L, d, increments):
    lt = M * (M + 1) // 2
    s = 1.0 / np.sqrt(d)                  # distances of order one whatever the width
    Z = rng.standard_normal((lt, T, 2, d) if increments else (lt, T, d)) * s
    X = np.cumsum(rng.standard_normal((N, L, d)) * 0.5 * s, axis=1)
    return Z, X


@pytest.mark.parametrize("M,T,N,L,d", [(4, 70, 9, 13, 12), (4, 33, 21, 7, 16), (4, 130, 6, 9, 28), (3, 20, 5, 11, 46), (4, 65, 4, 17, 126), (1, 5, 3, 4, 3),
                                       (2, 64, 7, 1, 8), (5, 12, 9, 6, 10), (8, 9, 3, 10, 14), (4, 7, 66, 5, 300)])
@pytest.mark.parametrize("base", WIDE_BASES)
def test_wide_tensor_vs_sequence_levels_and_gradient(M, T, N, L, d, base):
    """gpsig_tens_vs_seq_levels / _grad on the wide route (forced: option wide = 1, so that widths the tile kernel serves are covered too): 1 .. 8
    levels, ragged tensor counts across the 64-lane blocks, a single observation, differences on / off, increments on / off, the argument array in
    one chunk and in several (wide_chunk_mb = 1)."""
    if base != "rbf" and (M, d) in ((1, 3), (2, 8), (8, 14), (4, 300)):
        pytest.skip("a sample of the shapes is enough for the Matern families")
    rng = np.random.default_rng(1000 * M + T + d)
    ctx = _host_ctx()
    ctx.set_option("wide", 1)
    try:
        for difference in (True, False):
            for increments in (False, True):
                if L == 1 and difference:
                    continue
                Z, X = _data(rng, M, T, N, L, d, increments)
                G = rng.standard_normal((M + 1, T, N))
                kt = OT.SignatureKernelTorchOracle(d, M, base, difference=difference)
                tZ, tX = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True)
                want = kt.K_tens_vs_seq_levels(tZ, tX, increments)
                (want * torch.tensor(G)).sum().backward()
                keep = []
                p = _params(base, d, M, difference, keep)
                for mb in (0, 1):
                    ctx.set_option("wide_chunk_mb", mb)
                    out = np.full((M + 1, T, N), np.nan)
                    ctx.call("gpsig_tens_vs_seq_levels", p, _vp(Z), _vp(X), T, N, L, int(increments), _vp(out))
                    assert rel(out, want) < 1e-10, (difference, increments, mb, rel(out, want))
                    gZ, gX, gb = np.full_like(Z, np.nan), np.full_like(X, np.nan), np.zeros(2)
                    ctx.call("gpsig_tens_vs_seq_levels_grad", p, _vp(Z), _vp(X), T, N, L, int(increments), _vp(G), _vp(gZ), _vp(gX), gb.ctypes.data_as(_P))
                    assert rel(gZ, tZ.grad) < 1e-9 and rel(gX, tX.grad) < 1e-9, (difference, increments, mb, rel(gZ, tZ.grad), rel(gX, tX.grad))
    finally:
        ctx.set_option("wide", -1)
        ctx.set_option("wide_chunk_mb", 0)


@pytest.mark.parametrize("base,M,T,N,L,d", [("rbf", 4, 70, 45, 9, 12), ("matern32", 3, 40, 12, 6, 28), ("rbf", 4, 130, 9, 7, 126), ("matern52", 2, 10, 8, 6, 46),
                                            ("rbf", 4, 66, 10, 8, 6)])
def test_wide_weighted_sum_and_gradient(base, M, T, N, L, d):
    """gpsig_tens_vs_seq_weighted / _grad on the wide route: the level sum inside the kernel, the chain totals handed from the forward to the
    reverse call (device pointers) or rebuilt by it, gradients with respect to Z, X and the per-sequence factors."""
    from gpsig_amd import _lib
    rng = np.random.default_rng(7 * M + T + d)
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)               # (a stream of its own: the context of the default stream is the host-pointer one of _host_ctx)
    dctx = _lib.context(0, side.cuda_stream)
    dctx.set_pointer_mode(_lib.PTR_DEVICE)
    hctx = _host_ctx()
    ptr = lambda t_: C.c_void_p(t_.data_ptr())      # noqa: E731
    for ctx_ in (dctx, hctx):
        ctx_.set_option("wide", 1)
    try:
        for increments in (False, True):
            Z, X = _data(rng, M, T, N, L, d, increments)
            F = rng.uniform(0.5, 1.5, (N, M + 1))
            G = rng.standard_normal((T, N))
            kt = OT.SignatureKernelTorchOracle(d, M, base)
            tZ, tX, tF = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True), torch.tensor(F, requires_grad=True)
            want = (kt.K_tens_vs_seq_levels(tZ, tX, increments) * tF.t()[:, None, :]).sum(0)
            (want * torch.tensor(G)).sum().backward()
            keep = []
            p = _params(base, d, M, True, keep)
            out = np.empty((T, N))
            hctx.call("gpsig_tens_vs_seq_weighted", p, _vp(Z), _vp(X), T, N, L, int(increments), _vp(F), _vp(out), None, None)
            assert rel(out, want) < 1e-10
            gZ, gX, gF, gb = np.empty_like(Z), np.empty_like(X), np.empty_like(F), np.zeros(2)
            hctx.call("gpsig_tens_vs_seq_weighted_grad", p, _vp(Z), _vp(X), T, N, L, int(increments), _vp(F), _vp(G), None, _vp(gZ), _vp(gX), _vp(gF),
                      gb.ctypes.data_as(_P))
            assert rel(gZ, tZ.grad) < 1e-9 and rel(gX, tX.grad) < 1e-9 and rel(gF, tF.grad) < 1e-9, (increments, rel(gZ, tZ.grad), rel(gX, tX.grad), rel(gF, tF.grad))
            dZ, dX, dF, dG = (torch.tensor(a, device=dev) for a in (Z, X, F, G))
            dgZ, dgX, dgF, dgb = torch.empty_like(dZ), torch.empty_like(dX), torch.empty_like(dF), torch.zeros(2, dtype=torch.float64, device=dev)
            aux = torch.empty(int(_lib.load().gpsig_tens_vs_seq_aux_elems(C.byref(p), T, N)), dtype=torch.float64, device=dev)
            dout, wrote = torch.empty((T, N), dtype=torch.float64, device=dev), C.c_int32(0)
            torch.cuda.synchronize()
            dctx.call("gpsig_tens_vs_seq_weighted", p, ptr(dZ), ptr(dX), T, N, L, int(increments), ptr(dF), ptr(dout), ptr(aux), C.byref(wrote))
            side.synchronize()
            assert rel(dout, want) < 1e-10 and wrote.value == 1
            for use_aux in (False, True):
                dctx.call("gpsig_tens_vs_seq_weighted_grad", p, ptr(dZ), ptr(dX), T, N, L, int(increments), ptr(dF), ptr(dG), ptr(aux) if use_aux else None,
                          ptr(dgZ), ptr(dgX), ptr(dgF), C.cast(dgb.data_ptr(), _P))
                side.synchronize()
                assert rel(dgZ, tZ.grad) < 1e-9 and rel(dgX, tX.grad) < 1e-9 and rel(dgF, tF.grad) < 1e-9, (use_aux, rel(dgZ, tZ.grad), rel(dgX, tX.grad))
    finally:
        for ctx_ in (dctx, hctx):
            ctx_.set_option("wide", -1)


@pytest.mark.parametrize("base,d,num_lags,L,increments", [("rbf", 14, 1, 9, True), ("rbf", 6, 1, 12, True), ("matern32", 23, 1, 7, False), ("rbf", 63, 1, 11, True),
                                                          ("matern12", 16, 0, 8, True), ("rbf", 200, 0, 6, True)])
def test_wide_evaluation_path_against_the_oracle(base, d, num_lags, L, increments):
    """kernels.Signature*.K_tens_vs_seq / K_tens_n_seq_covs (the fused evaluation entry points: scaling by lengthscales, lags, normalisation, weights) at
    the reference's own settings -- time-augmented data with num_lags = 1: 2 d columns -- against the NumPy oracle."""
    from gpsig_amd import kernels
    import test_gpu_parity as P
    rng = np.random.default_rng(31 + d)
    M, T, N = 4, 70, 9
    lt = M * (M + 1) // 2
    de = d * (num_lags + 1)
    X = np.cumsum(rng.standard_normal((N, L, d)) * 0.4, axis=1)
    Z = rng.standard_normal((lt, T, 2, de) if increments else (lt, T, de)) * 0.7
    ls = rng.uniform(0.8, 1.6, d) * np.sqrt(d)
    var = rng.uniform(0.5, 1.5, M + 1)
    # (Matern-1/2 normalised: kappa(x, x) = exp(-sqrt(max(rounding noise, 1e-40))) is 1e-8 from one in the reference's own arithmetic -- DESIGN section 5)
    tol = 1e-7 if base == "matern12" else 1e-10
    for normalization in (True, False):
        kw = dict(base=base, input_dim=L * d, num_features=d, num_levels=M, lengthscales=ls, variances=var, normalization=normalization,
                  num_lags=num_lags or None)
        k, ko = P.make_kernel(kernels, kw), P.make_oracle(kw)
        if num_lags:
            k.lags = ko.lags = np.array([0.23])
            k.gamma = ko.gamma = np.array([0.6, 0.45])
        got = k.K_tens_vs_seq(Z, X.reshape(N, -1), increments=increments)
        want = ko.K_tens_vs_seq(Z, X.reshape(N, -1), increments=increments)
        assert rel(got, want) < tol, (normalization, rel(got, want))
        gl = k.K_tens_vs_seq(Z, X.reshape(N, -1), increments=increments, return_levels=True)
        wl = ko.K_tens_vs_seq(Z, X.reshape(N, -1), increments=increments, return_levels=True)
        assert max(rel(a, b) for a, b in zip(gl[1:], wl[1:])) < tol
        for a, b in zip(k.K_tens_n_seq_covs(Z, X.reshape(N, -1), increments=increments), ko.K_tens_n_seq_covs(Z, X.reshape(N, -1), increments=increments)):
            assert rel(a, b) < tol


@pytest.mark.parametrize("base,d,num_lags", [("rbf", 14, 1), ("rbf", 63, 1), ("matern32", 23, 1), ("rbf", 150, 0)])
def test_wide_module_gradients(base, d, num_lags):
    """autodiff.SignatureKernelModule.K_tens_n_seq_covs at the reference's settings (increments, num_lags = 1): values and the gradients with respect to
    the inducing tensors, lengthscales, lags, lag weights and variances against autograd of the differentiable oracle."""
    from gpsig_amd import autodiff, kernels
    import test_gpu_parity as P
    rng = np.random.default_rng(5 + d)
    M, T, N, L = 4, 40, 7, 9
    lt = M * (M + 1) // 2
    de = d * (num_lags + 1)
    X = np.cumsum(rng.standard_normal((N, L, d)) * 0.4, axis=1).reshape(N, -1)
    Z = rng.standard_normal((lt, T, 2, de)) * 0.7
    ls = rng.uniform(0.8, 1.6, d) * np.sqrt(d)
    kw = dict(base=base, input_dim=L * d, num_features=d, num_levels=M, lengthscales=ls, num_lags=num_lags or None)
    kern = P.make_kernel(kernels, kw)
    mod = autodiff.SignatureKernelModule(kern, device="cuda:0")
    Zg = torch.tensor(Z, device="cuda:0", requires_grad=True)
    Wt = [rng.standard_normal(s) for s in ((T, T), (T, N), (N,))]
    outs = mod.K_tens_n_seq_covs(Zg, torch.tensor(X, device="cuda:0"), increments=True)
    sum((o * torch.tensor(w, device="cuda:0")).sum() for o, w in zip(outs, Wt)).backward()
    lsr = torch.tensor(ls, requires_grad=True)
    okw = dict(lengthscales=lsr)
    if num_lags:
        lagr, gamr = mod.lags.detach().cpu().clone().requires_grad_(True), mod.gamma.detach().cpu().clone().requires_grad_(True)
        okw.update(num_lags=num_lags, lags=lagr, gamma=gamr)
    orc = OT.SignatureKernelTorchOracle(d, M, base, **okw)
    Zc = torch.tensor(Z, requires_grad=True)
    wants = orc.K_tens_n_seq_covs(Zc, torch.tensor(X), increments=True)
    sum((o * torch.tensor(w)).sum() for o, w in zip(wants, Wt)).backward()
    for o, w in zip(outs, wants):
        assert rel(o, w) < 1e-10
    assert rel(Zg.grad, Zc.grad) < 1e-8
    sig = lambda r: torch.sigmoid(r.detach().cpu())      # noqa: E731
    assert rel(mod.raw_lengthscales.grad, lsr.grad * sig(mod.raw_lengthscales)) < 1e-8
    if num_lags:
        lg = mod.lags.detach().cpu()
        assert rel(mod.raw_lags.grad, lagr.grad * lg * (1 - lg)) < 1e-8
        assert rel(mod.raw_gamma.grad, gamr.grad * sig(mod.raw_gamma)) < 1e-8


@pytest.mark.parametrize("M,N1,N2,L1,L2,d,kind", [(4, 7, 5, 9, 13, 12, "cross"), (3, 6, 6, 70, 70, 28, "sym"), (4, 9, 9, 33, 33, 46, "diag"), (4, 3, 4, 20, 131, 126, "cross"),
                                                  (5, 2, 2, 300, 300, 3, "diag"), (2, 3, 2, 40, 260, 10, "cross"), (7, 3, 3, 12, 12, 14, "sym"), (1, 4, 3, 5, 6, 300, "cross"), (4, 37, 37, 21, 21, 10, "sym"),
                                                  (4, 70, 70, 8, 8, 16, "diag"), (4, 5, 5, 93, 93, 28, "diag")])
@pytest.mark.parametrize("base", WIDE_BASES)
def test_wide_sequence_lattices_and_gradient(M, N1, N2, L1, L2, d, kind, base):
    """gpsig_seq_gram_levels / gpsig_seq_diag_levels and their gradients on the wide route (forced: option wide = 1): the argument lattices by dgemm
    (one product of all points, or one per sequence for the diagonal), one wavefront per lattice with 1 / 2 / 4 / 8 columns per lane (up to 64 / 128 /
    256 / 512 lattice columns), 1 .. 7 levels, differences on / off, the lattices in one chunk and in several."""
    if base != "rbf" and (M, d) in ((5, 3), (2, 10), (7, 14), (1, 300), (4, 16)):
        pytest.skip("a sample of the shapes is enough for the Matern families")
    rng = np.random.default_rng(100 * M + L2 + d)
    ctx = _host_ctx()
    ctx.set_option("wide", 1)
    s = 1.0 / np.sqrt(d)
    try:
        for difference in (True, False):
            X = np.cumsum(rng.standard_normal((N1, L1, d)) * 0.5 * s, axis=1)
            Y = np.cumsum(rng.standard_normal((N2, L2, d)) * 0.5 * s, axis=1) if kind == "cross" else None
            G = rng.standard_normal((M + 1, N1) if kind == "diag" else (M + 1, N1, N2 if kind == "cross" else N1))
            kt = OT.SignatureKernelTorchOracle(d, M, base, difference=difference)
            tX = torch.tensor(X, requires_grad=True)
            tY = None if Y is None else torch.tensor(Y, requires_grad=True)
            want = kt.K_seq_diag_levels(tX) if kind == "diag" else kt.K_seq_levels(tX, tY)
            (want * torch.tensor(G)).sum().backward()
            keep = []
            p = _params(base, d, M, difference, keep)
            # (waves: wavefronts per lattice -- -1 the planner's choice: the lattice's columns over several wavefronts of one column per lane where a
            # launch holds few long lattices; 0: one wavefront per lattice with 1 / 2 / 4 / 8 columns per lane; 1: 2 / 4 / 8 wavefronts instead wherever possible)
            # fold: the symmetric Gram's reverse pass over the pairs i <= j with the upstream gradient folded onto them (1, the default) or over all ordered pairs
            # o1: lattices of at most 64 columns four to a wavefront from a dM lattice (seq_grad_wave_o1_kernel: 2 = wherever the shape fits; the planner takes it from
            # 1,024 lattices), 0 = the lattice kernels
            for mb, waves, fold, o1 in ((0, -1, 1, 1), (1, -1, 1, 2), (0, 0, 1, 0), (0, 1, 1, 1), (0, -1, 1, 2)) + (((0, -1, 0, 2), (0, -1, 0, 0)) if kind == "sym" else ()):
                ctx.set_option("wide_chunk_mb", mb)
                ctx.set_option("wide_lat_waves", waves)
                ctx.set_option("wide_sym_fold", fold)
                ctx.set_option("wide_o1_sweeps", o1)
                out = np.full(G.shape, np.nan)
                gX, gY, gb = np.full_like(X, np.nan), (None if Y is None else np.full_like(Y, np.nan)), np.zeros(2)
                if kind == "diag":
                    ctx.call("gpsig_seq_diag_levels", p, _vp(X), N1, L1, _vp(out))
                    ctx.call("gpsig_seq_diag_levels_grad", p, _vp(X), N1, L1, _vp(G), _vp(gX), gb.ctypes.data_as(_P))
                else:
                    n2, l2 = (N2, L2) if Y is not None else (N1, L1)
                    ctx.call("gpsig_seq_gram_levels", p, _vp(X), _vp(Y), N1, n2, L1, l2, _vp(out))
                    ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N1, n2, L1, l2, _vp(G), _vp(gX), _vp(gY), gb.ctypes.data_as(_P))
                # (matern12: a sequence against itself has coinciding points -- the float64 oracle's kappa there is exp(-sqrt(rounding noise)), 1e-8 from
                # one, its derivative whatever the noise makes of 1 / r; the product takes such distances as zero: DESIGN section 5)
                tv, tg = (1e-6, 1e-5) if (base == "matern12" and kind != "cross") else (1e-9, 1e-8)
                assert rel(out, want) < tv, (difference, mb, waves, o1, rel(out, want))
                assert rel(gX, tX.grad) < tg, (difference, mb, waves, o1, rel(gX, tX.grad))
                if Y is not None:
                    assert rel(gY, tY.grad) < 1e-8, (difference, mb, waves, o1, rel(gY, tY.grad))
    finally:
        ctx.set_option("wide", -1)
        ctx.set_option("wide_chunk_mb", 0)
        ctx.set_option("wide_lat_waves", -1)
        ctx.set_option("wide_sym_fold", 1)
        ctx.set_option("wide_o1_sweeps", 1)


def test_wide_route_is_what_the_reference_shapes_take():
    """The settings of benchmarks/run_gpsig_benchmarks.py:32 on the data sets of benchmarks/datasets.json whose state space has more than 8 columns
    (NetFlow / Wafer / ArabicDigits / AUSLAN / CMUsubject16 / PEMS: 10 / 14 / 28 / 46 / 126 / 1,928): the library's timing record names the wide
    kernels for Kzx and -- where the exact-shape pair kernels are not built (more than 32 columns, more than 256 observations beyond 8 columns) --
    for the level diagonals: no one-pair-per-thread fallback, no older mapping."""
    from gpsig_amd import _lib, kernels
    rng = np.random.default_rng(3)
    M, T, N = 4, 64, 6
    lt = M * (M + 1) // 2
    dev = torch.device("cuda:0")
    for d, L, diag_wide in ((5, 300, True), (7, 60, False), (14, 30, False), (23, 40, True), (63, 50, True), (964, 12, True)):
        kern = kernels.SignatureRBF(L * d, d, M, lengthscales=np.sqrt(d), num_lags=1, normalization=False)
        X = torch.tensor(np.cumsum(rng.standard_normal((N, L, d)) * 0.3, axis=1).reshape(N, -1), device=dev)
        Z = torch.tensor(rng.standard_normal((lt, T, 2, 2 * d)), device=dev)
        ctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
        for call, name, expected in ((lambda: kern.K_tens_vs_seq(Z, X, increments=True), "wide_tvs", True),
                                     (lambda: kern.Kdiag(X, return_levels=True), "wide_lattice", diag_wide)):
            ctx.timing_reset()
            call()
            got = ctx.timing_info()[0]
            assert (name in str(got)) == expected, (d, L, name, got)


@pytest.mark.parametrize("M,T,d", [(4, 70, 14), (4, 130, 46), (3, 33, 126), (1, 5, 3), (6, 12, 28), (8, 9, 10), (4, 64, 300)])
@pytest.mark.parametrize("base", WIDE_BASES)
def test_wide_tensor_gram_levels_and_gradient(M, T, d, base):
    """gpsig_tens_gram_levels / _grad on the wide route (forced): the argument blocks of every component by one batched dgemm (left-form rows times
    right-form rows of the same tensors), the four-term difference of kernels.py:276-277, the level products of signature_algs.py:91-97 and their
    reverse pass; 1 .. 8 levels, ragged tensor counts, increments on / off."""
    if base != "rbf" and (M, d) in ((1, 3), (8, 10), (4, 300)):
        pytest.skip("a sample of the shapes is enough for the Matern families")
    rng = np.random.default_rng(10 * M + T + d)
    ctx = _host_ctx()
    ctx.set_option("wide", 1)
    lt = M * (M + 1) // 2
    try:
        for increments in (False, True):
            Z = rng.standard_normal((lt, T, 2, d) if increments else (lt, T, d)) / np.sqrt(d)
            G = rng.standard_normal((M + 1, T, T))
            kt = OT.SignatureKernelTorchOracle(d, M, base)
            tZ = torch.tensor(Z, requires_grad=True)
            want = kt.K_tens_levels(tZ, increments)
            (want * torch.tensor(G)).sum().backward()
            keep = []
            p = _params(base, d, M, True, keep)
            out, gZ, gb = np.full((M + 1, T, T), np.nan), np.full_like(Z, np.nan), np.zeros(2)
            ctx.call("gpsig_tens_gram_levels", p, _vp(Z), T, int(increments), _vp(out))
            ctx.call("gpsig_tens_gram_levels_grad", p, _vp(Z), T, int(increments), _vp(G), _vp(gZ), gb.ctypes.data_as(_P))
            if base == "matern12":          # a tensor against itself: coinciding points (the clamp of kernels.py:781 in both; values 1e-8 from one)
                off = ~np.eye(T, dtype=bool)
                assert rel(out[:, off], want.detach().numpy()[:, off]) < 1e-9
            else:
                assert rel(out, want) < 1e-10, (increments, rel(out, want))
            assert rel(gZ, tZ.grad) < (1e-5 if base == "matern12" else 1e-8), (increments, rel(gZ, tZ.grad))      # (matern12: the oracle's 1 / r at rounding-noise distances)
    finally:
        ctx.set_option("wide", -1)


@pytest.mark.parametrize("M,order,T,N,L,d", [(4, 2, 70, 9, 13, 12), (4, 4, 33, 6, 9, 28), (3, 2, 20, 5, 11, 6), (5, 3, 40, 4, 8, 46), (6, 4, 12, 5, 7, 10), (2, 2, 64, 7, 5, 3), (8, 2, 9, 3, 6, 14),
                                            (4, 3, 70, 9, 13, 4), (5, 4, 33, 6, 9, 8), (4, 2, 140, 17, 12, 6), (5, 2, 20, 5, 10, 5), (3, 3, 65, 70, 6, 7)])
@pytest.mark.parametrize("base", ["rbf", "matern32"])
def test_wide_higher_order_chains_and_gradient(M, order, T, N, L, d, base):
    """Round 6: the higher-order tensor-vs-sequence chains (signature_algs.py:129-160, order <= 4) on the wide route: values, and the reverse pass that
    rebuilds a step's repeat-count vectors from the chain totals -- levels and the weighted sum, against autograd of the oracle; at <= 8 columns too,
    where SignatureRBF with 3-5 levels runs the tile kernels' higher-order instances (tvs_tile_inst_ho.hip forward, tvs_grad_tile_inst_ho.hip reverse, continuing
    from the forward's chain totals; the Matern families the reverse instances) unless the wide route is forced: both are checked."""
    if base != "rbf" and ((M, order) in ((2, 2), (8, 2), (6, 4)) or d in (5, 7)):
        pytest.skip("a sample of the shapes is enough for the Matern families")
    from gpsig_amd import _lib
    rng = np.random.default_rng(10 * M + order + d)
    ctx = _host_ctx()
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    dctx = _lib.context(0, side.cuda_stream)
    dctx.set_pointer_mode(_lib.PTR_DEVICE)
    ptr = lambda t_: C.c_void_p(t_.data_ptr())      # noqa: E731
    from gpsig_amd.autodiff import _Spec
    for increments in (False, True):
        for difference in (True, False):
            Z, X = _data(rng, M, T, N, L, d, increments)
            G = rng.standard_normal((M + 1, T, N))
            kt = OT.SignatureKernelTorchOracle(d, M, base, difference=difference, order=order)
            tZ, tX = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True)
            want = kt.K_tens_vs_seq_levels(tZ, tX, increments)
            (want * torch.tensor(G)).sum().backward()
            keep = []
            p = _Spec(base, M, difference, 0.0, order=order).params(d, 0.0, keep)
            for wide in ((1, -1) if d <= 8 else (-1,)):
                ctx.set_option("wide", wide)
                try:
                    out = np.full((M + 1, T, N), np.nan)
                    ctx.call("gpsig_tens_vs_seq_levels", p, _vp(Z), _vp(X), T, N, L, int(increments), _vp(out))
                    gZ, gX, gb = np.full_like(Z, np.nan), np.full_like(X, np.nan), np.zeros(2)
                    ctx.call("gpsig_tens_vs_seq_levels_grad", p, _vp(Z), _vp(X), T, N, L, int(increments), _vp(G), _vp(gZ), _vp(gX), gb.ctypes.data_as(_P))
                finally:
                    ctx.set_option("wide", -1)
                assert rel(out, want) < 1e-10, (increments, difference, wide, rel(out, want))
                assert rel(gZ, tZ.grad) < 1e-9 and rel(gX, tX.grad) < 1e-9, (increments, difference, wide, rel(gZ, tZ.grad), rel(gX, tX.grad))
        # the weighted sum with the chain totals handed over (device pointers)
        Z, X = _data(rng, M, T, N, L, d, increments)
        F, G2 = rng.uniform(0.5, 1.5, (N, M + 1)), rng.standard_normal((T, N))
        kt = OT.SignatureKernelTorchOracle(d, M, base, order=order)
        tZ, tX, tF = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True), torch.tensor(F, requires_grad=True)
        want = (kt.K_tens_vs_seq_levels(tZ, tX, increments) * tF.t()[:, None, :]).sum(0)
        (want * torch.tensor(G2)).sum().backward()
        keep = []
        p = _Spec(base, M, True, 0.0, order=order).params(d, 0.0, keep)
        dZ, dX, dF, dG = (torch.tensor(a, device=dev) for a in (Z, X, F, G2))
        dgZ, dgX, dgF, dgb = torch.empty_like(dZ), torch.empty_like(dX), torch.empty_like(dF), torch.zeros(2, dtype=torch.float64, device=dev)
        aux = torch.empty(int(_lib.load().gpsig_tens_vs_seq_aux_elems(C.byref(p), T, N)), dtype=torch.float64, device=dev)
        dout, wrote = torch.empty((T, N), dtype=torch.float64, device=dev), C.c_int32(0)
        torch.cuda.synchronize()
        dctx.call("gpsig_tens_vs_seq_weighted", p, ptr(dZ), ptr(dX), T, N, L, int(increments), ptr(dF), ptr(dout), ptr(aux), C.byref(wrote))
        side.synchronize()
        assert rel(dout, want) < 1e-10
        for use_aux in ((False, True) if wrote.value else (False,)):
            dctx.call("gpsig_tens_vs_seq_weighted_grad", p, ptr(dZ), ptr(dX), T, N, L, int(increments), ptr(dF), ptr(dG), ptr(aux) if use_aux else None,
                      ptr(dgZ), ptr(dgX), ptr(dgF), C.cast(dgb.data_ptr(), _P))
            side.synchronize()
            assert rel(dgZ, tZ.grad) < 1e-9 and rel(dgX, tX.grad) < 1e-9 and rel(dgF, tF.grad) < 1e-9, (use_aux, rel(dgZ, tZ.grad), rel(dgX, tX.grad), rel(dgF, tF.grad))


@pytest.mark.parametrize("M,order,N1,N2,L1,L2,d,kind", [(4, 2, 7, 5, 9, 13, 12, "cross"), (3, 3, 6, 6, 40, 40, 28, "sym"), (4, 2, 9, 9, 33, 33, 70, "diag"), (5, 4, 3, 4, 20, 61, 126, "cross"),
                                                       (4, 3, 5, 5, 100, 100, 5, "diag"), (2, 2, 3, 2, 40, 260, 10, "cross"), (5, 2, 20, 20, 8, 8, 16, "sym"), (4, 4, 4, 4, 12, 12, 300, "sym")])
@pytest.mark.parametrize("base", ["rbf", "matern32", "matern12"])
def test_wide_higher_order_lattices_and_gradient(M, order, N1, N2, L1, L2, d, kind, base):
    """Round 6: the higher-order sequence recursion (signature_algs.py:37-74) on the wide route, both directions -- argument lattices by dgemm, dM by
    wide_lattice_dm_kernel, one forward sweep per pair (seq_levels_wave_ho_kernel: K_m < M from the 2-D prefixes at the last cell, K_M from the cells'
    level-M grids) or the two sweeps of the reverse pass, the adjoint contracted back -- at any number of columns: values and gradients against the
    oracle's autograd, forced (option wide = 1) and as the planner routes it, in one chunk and in several."""
    if base != "rbf" and (M, order) in ((2, 2), (5, 2), (4, 4), (5, 4)):
        pytest.skip("a sample of the shapes is enough for the Matern families")
    rng = np.random.default_rng(100 * M + L2 + d + order)
    ctx = _host_ctx()
    s = 1.0 / np.sqrt(d)
    try:
        for difference in (True, False):
            X = np.cumsum(rng.standard_normal((N1, L1, d)) * 0.5 * s, axis=1)
            Y = np.cumsum(rng.standard_normal((N2, L2, d)) * 0.5 * s, axis=1) if kind == "cross" else None
            G = rng.standard_normal((M + 1, N1) if kind == "diag" else (M + 1, N1, N2 if kind == "cross" else N1)) * (1.0 if difference else 1e-2)
            kt = OT.SignatureKernelTorchOracle(d, M, base, difference=difference, order=order)
            tX = torch.tensor(X, requires_grad=True)
            tY = None if Y is None else torch.tensor(Y, requires_grad=True)
            want = kt.K_seq_diag_levels(tX) if kind == "diag" else kt.K_seq_levels(tX, tY)
            (want * torch.tensor(G)).sum().backward()
            keep = []
            from gpsig_amd.autodiff import _Spec
            p = _Spec(base, M, difference, 0.0, order=order).params(d, 0.0, keep)
            for wide, mb in ((1, 0), (1, 1), (-1, 0)):
                ctx.set_option("wide", wide)
                ctx.set_option("wide_chunk_mb", mb)
                out = np.full(G.shape, np.nan)
                gX, gY, gb = np.full_like(X, np.nan), (None if Y is None else np.full_like(Y, np.nan)), np.zeros(2)
                if kind == "diag":
                    ctx.call("gpsig_seq_diag_levels", p, _vp(X), N1, L1, _vp(out))
                    ctx.call("gpsig_seq_diag_levels_grad", p, _vp(X), N1, L1, _vp(G), _vp(gX), gb.ctypes.data_as(_P))
                else:
                    n2, l2 = (N2, L2) if Y is not None else (N1, L1)
                    ctx.call("gpsig_seq_gram_levels", p, _vp(X), _vp(Y), N1, n2, L1, l2, _vp(out))
                    ctx.call("gpsig_seq_gram_levels_grad", p, _vp(X), _vp(Y), N1, n2, L1, l2, _vp(G), _vp(gX), _vp(gY), gb.ctypes.data_as(_P))
                tv, tg = (1e-6, 1e-5) if (base == "matern12" and kind != "cross") else (1e-9, 1e-8)
                assert rel(out, want) < tv, (difference, wide, mb, rel(out, want))
                assert rel(gX, tX.grad) < tg, (difference, wide, mb, rel(gX, tX.grad))
                if Y is not None:
                    assert rel(gY, tY.grad) < 1e-8, (difference, wide, mb, rel(gY, tY.grad))
    finally:
        ctx.set_option("wide", -1)
        ctx.set_option("wide_chunk_mb", 0)


@pytest.mark.parametrize("d,num_lags,order", [(6, 1, 1), (13, 1, 1), (5, 1, 2)])
def test_wide_training_step_recorded_as_one_hip_graph(d, num_lags, order):
    """SVGPModule.fit(graph=True) at the reference's kind of shape (time-augmented columns doubled by one lag: 12 / 26 / 10 columns, inducing tensors with
    increments): the wide route's dgemms and kernels record into the step's HIP graph like the exact-shape kernels -- the replayed steps give the eager
    loop's ELBO trace and parameters."""
    from gpsig_amd import kernels, models, likelihoods as LK, inducing_variables as iv
    rng = np.random.default_rng(48)
    N, L, M, T = 30, 15, 4, 9
    X = np.cumsum(rng.standard_normal((N, L, d)) * 0.3 / np.sqrt(d), axis=1).reshape(N, -1)
    Y = rng.integers(0, 2, (N, 1)).astype(np.float64)
    Xg, Yg = torch.tensor(X, device="cuda:0"), torch.tensor(Y, device="cuda:0")
    Z = rng.standard_normal((M * (M + 1) // 2, T, 2, d * (num_lags + 1))) * 0.4 / np.sqrt(d)
    out = {}
    for graph in (False, True):
        kern = kernels.SignatureRBF(L * d, d, M, lengthscales=np.sqrt(d), num_lags=num_lags, order=order)
        model = models.SVGPModule(kern, iv.InducingTensors(Z.copy(), M, increments=True), LK.Bernoulli(), num_data=N, device="cuda:0")
        trace = model.fit(Xg, Yg, iterations=10, lr=0.02, minibatch_size=12, seed=5, graph=graph)
        out[graph] = (np.asarray(trace), [p.detach().cpu().numpy().copy() for p in model.parameters()])
    assert len(out[True][0]) == 10 and np.isfinite(out[True][0]).all()
    assert np.abs(out[True][0] - out[False][0]).max() <= 1e-5 * np.abs(out[False][0]).max()
    for a, b in zip(out[True][1], out[False][1]):
        assert np.abs(a - b).max() <= 1e-4 * (np.abs(b).max() + 1e-12)
