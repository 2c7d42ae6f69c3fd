"""CPU checks of the gradient path's building blocks.

1. The differentiable oracle (oracle/sigkern_oracle_torch.py) is pinned: its values equal the NumPy oracle's, and
   its autograd gradients equal central finite differences of the NumPy oracle.
2. The per-pair adjoint code the gfx950 gradient kernels run (gpsig_amd/csrc/grad_core.hpp, compiled for the host by
   tests/emu/emu_grad.cpp) reproduces torch.autograd of that oracle for every base kernel and layout.
"""
import numpy as np
import pytest
import torch

from oracle import sigkern_oracle as O
from oracle import sigkern_oracle_torch as OT
from tests import emu_grad_util as EG

BASES = ["linear", "rbf", "cosine", "poly", "mix", "matern12", "matern32", "matern52"]


def _bp(base):
    return (1.0, 3.0) if base == "poly" else ((0.4, 0.0) if base == "mix" else (0.0, 0.0))


def _np_kern(base, d, M, **kw):
    p0, p1 = _bp(base)
    bp = {"gamma": p0, "degree": p1} if base == "poly" else ({"mixing": p0} if base == "mix" else None)
    return O.SignatureKernelOracle(d * 1, d, M, base=base, base_params=bp, **kw)


def _t_kern(base, d, M, **kw):
    p0, p1 = _bp(base)
    return OT.SignatureKernelTorchOracle(d, M, base, p0=torch.tensor(p0, dtype=torch.float64, requires_grad=True), p1=p1, **kw)


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-300))


@pytest.mark.parametrize("base", BASES)
@pytest.mark.parametrize("normalization", [True, False])
def test_torch_oracle_values_equal_numpy_oracle(base, normalization):
    rng = np.random.default_rng(3)
    N, N2, L, d, M, T = 4, 3, 6, 3, 3, 4
    X, X2 = rng.standard_normal((N, L * d)) * 0.6, rng.standard_normal((N2, L * d)) * 0.6
    ls = rng.uniform(0.7, 1.5, d)
    var = rng.uniform(0.5, 1.5, M + 1)
    kn = _np_kern(base, d, M, normalization=normalization, lengthscales=ls, variances=var)
    kn.input_dim = L * d
    kt = _t_kern(base, d, M, normalization=normalization, lengthscales=ls, variances=var)
    tX, tX2 = torch.tensor(X), torch.tensor(X2)
    tol = 1e-6 if base == "matern12" else 1e-11
    assert rel(kt.K(tX).detach(), kn.K(X)) < tol
    assert rel(kt.K(tX, tX2).detach(), kn.K(X, X2)) < tol
    assert rel(kt.Kdiag(tX).detach(), kn.Kdiag(X)) < tol
    for incr in (False, True):
        Z = rng.standard_normal((M * (M + 1) // 2, T, 2, d) if incr else (M * (M + 1) // 2, T, d))
        tZ = torch.tensor(Z)
        assert rel(kt.K_tens(tZ, increments=incr).detach(), kn.K_tens(Z, increments=incr)) < tol
        assert rel(kt.K_tens_vs_seq(tZ, tX, increments=incr).detach(), kn.K_tens_vs_seq(Z, X, increments=incr)) < tol
        a = kt.K_tens_n_seq_covs(tZ, tX, increments=incr)
        b = kn.K_tens_n_seq_covs(Z, X, increments=incr)
        for u, v in zip(a, b):
            assert rel(u.detach(), v) < tol


@pytest.mark.parametrize("base", ["linear", "rbf"])
@pytest.mark.parametrize("order", [2, 3, 4])
def test_torch_oracle_higher_order_equals_numpy_oracle(base, order):
    """signature_algs.py:37-74 and :129-160 restated in torch (the gradient oracle of the higher-order algorithms) against the
    NumPy oracle, which the notebook identities pin at order = num_levels (tests/test_oracle.py)."""
    rng = np.random.default_rng(11)
    N, N2, L, d, M, T = 4, 3, 7, 3, 4, 3
    X, X2 = rng.standard_normal((N, L * d)) * 0.6, rng.standard_normal((N2, L * d)) * 0.6
    ls, var = rng.uniform(0.7, 1.5, d), rng.uniform(0.5, 1.5, M + 1)
    for norm in (True, False):
        kn = _np_kern(base, d, M, normalization=norm, lengthscales=ls, variances=var, order=order)
        kn.input_dim = L * d
        kt = _t_kern(base, d, M, normalization=norm, lengthscales=ls, variances=var, order=order)
        tX, tX2 = torch.tensor(X), torch.tensor(X2)
        assert rel(kt.K(tX).detach(), kn.K(X)) < 1e-11
        assert rel(kt.K(tX, tX2, return_levels=True).detach(), kn.K(X, X2, return_levels=True)) < 1e-11
        assert rel(kt.Kdiag(tX).detach(), kn.Kdiag(X)) < 1e-11
        for incr in (False, True):
            Z = rng.standard_normal((M * (M + 1) // 2, T, 2, d) if incr else (M * (M + 1) // 2, T, d))
            assert rel(kt.K_tens_vs_seq(torch.tensor(Z), tX, increments=incr, return_levels=True).detach(),
                       kn.K_tens_vs_seq(Z, X, increments=incr, return_levels=True)) < 1e-11


def test_torch_oracle_with_lags_equals_numpy_oracle():
    rng = np.random.default_rng(4)
    N, L, d, M = 3, 7, 2, 3
    X = rng.standard_normal((N, L * d))
    kn = O.SignatureKernelOracle(L * d, d, M, base="rbf", num_lags=2, lengthscales=1.3)
    kt = OT.SignatureKernelTorchOracle(d, M, "rbf", lengthscales=kn.lengthscales, num_lags=2, lags=kn.lags, gamma=kn.gamma)
    assert rel(kt.K(torch.tensor(X)).detach(), kn.K(X)) < 1e-11


def test_torch_oracle_gradient_equals_finite_differences_of_numpy_oracle():
    rng = np.random.default_rng(5)
    N, L, d, M = 3, 5, 2, 3
    X = rng.standard_normal((N, L * d)) * 0.7
    W = rng.standard_normal((N, N))
    ls0, var0 = np.array([0.9, 1.4]), rng.uniform(0.5, 1.5, M + 1)

    def loss_np(ls, var):
        k = O.SignatureKernelOracle(L * d, d, M, base="rbf", lengthscales=ls, variances=var)
        return float((k.K(X) * W).sum())
    ls = torch.tensor(ls0, requires_grad=True)
    var = torch.tensor(var0, requires_grad=True)
    kt = OT.SignatureKernelTorchOracle(d, M, "rbf", lengthscales=ls, variances=var)
    (kt.K(torch.tensor(X)) * torch.tensor(W)).sum().backward()
    h = 1e-6
    for i in range(d):
        e = np.zeros(d); e[i] = h
        fd = (loss_np(ls0 + e, var0) - loss_np(ls0 - e, var0)) / (2 * h)
        assert abs(fd - ls.grad[i].item()) < 1e-6 * max(1.0, abs(fd))
    for i in range(M + 1):
        e = np.zeros(M + 1); e[i] = h
        fd = (loss_np(ls0, var0 + e) - loss_np(ls0, var0 - e)) / (2 * h)
        assert abs(fd - var.grad[i].item()) < 1e-6 * max(1.0, abs(fd))


@pytest.mark.parametrize("base", BASES)
@pytest.mark.parametrize("difference", [True, False])
def test_seq_pair_adjoint_equals_autograd(base, difference):
    rng = np.random.default_rng(11)
    p0, p1 = _bp(base)
    tol = 1e-6 if base == "matern12" else 1e-12
    for (M, N1, N2, L1, L2, d, kind) in [(4, 3, 2, 6, 5, 3, "cross"), (3, 3, 3, 5, 5, 2, "sym"), (5, 4, 4, 7, 7, 5, "diag"), (1, 2, 3, 4, 2, 9, "cross")]:
        X = rng.standard_normal((N1, L1, d)) * 0.5
        Y = rng.standard_normal((N2, L2, d)) * 0.5 if kind == "cross" else None
        G = rng.standard_normal((M + 1, N1) if kind == "diag" else (M + 1, N1, N2 if kind == "cross" else N1))
        kt = _t_kern(base, d, M, difference=difference)
        tX = torch.tensor(X, requires_grad=True)
        tY = None if Y is None else torch.tensor(Y, requires_grad=True)
        lev = kt.K_seq_diag_levels(tX) if kind == "diag" else kt.K_seq_levels(tX, tY)
        (lev * torch.tensor(G)).sum().backward()
        levels, gX, gY, gp0 = EG.seq_grad(X, Y, G, M, base, difference, p0, p1, diag=(kind == "diag"))
        assert rel(levels, lev.detach()) < tol
        assert rel(gX, tX.grad) < tol
        if Y is not None:
            assert rel(gY, tY.grad) < tol
        if base in ("poly", "mix"):
            assert abs(gp0 - kt.p0.grad.item()) < 1e-11 * max(1.0, abs(gp0))


@pytest.mark.parametrize("base", BASES)
@pytest.mark.parametrize("difference", [True, False])
@pytest.mark.parametrize("increments", [False, True])
def test_tensor_adjoints_equal_autograd(base, difference, increments):
    rng = np.random.default_rng(12)
    p0, p1 = _bp(base)
    M, T, N, L, d = 4, 3, 4, 6, 3
    lt = M * (M + 1) // 2
    Z = rng.standard_normal((lt, T, 2, d) if increments else (lt, T, d)) * 0.5
    X = rng.standard_normal((N, L, d)) * 0.5
    G = rng.standard_normal((M + 1, T, N))
    kt = _t_kern(base, d, M, difference=difference)
    tZ, tX = torch.tensor(Z, requires_grad=True), torch.tensor(X, requires_grad=True)
    lev = kt.K_tens_vs_seq_levels(tZ, tX, increments)
    (lev * torch.tensor(G)).sum().backward()
    for fused in (False, True):       # storage-based and scratch-free (undo) formulations
        levels, gZ, gX, gp0 = EG.tvs_grad(Z, X, G, M, base, difference, increments, p0, p1, fused=fused)
        tol = 1e-10 if fused else 1e-12
        assert rel(levels, lev.detach()) < 1e-12 and rel(gZ, tZ.grad) < tol and rel(gX, tX.grad) < tol, (fused, rel(gZ, tZ.grad), rel(gX, tX.grad))
        if base in ("poly", "mix"):
            assert abs(gp0 - kt.p0.grad.item()) < 1e-9 * max(1.0, abs(gp0))
    # tensor vs tensor
    G2 = rng.standard_normal((M + 1, T, T))
    kt = _t_kern(base, d, M)
    tZ = torch.tensor(Z, requires_grad=True)
    (kt.K_tens_levels(tZ, increments) * torch.tensor(G2)).sum().backward()
    for row_owned in (False, True):       # one (t, t') entry per thread / one tensor per thread using the symmetry of Kzz
        gZ, gp0 = EG.tens_grad(Z, G2, M, base, increments, p0, p1, row_owned=row_owned)
        assert rel(gZ, tZ.grad) < 1e-12, row_owned
        if base in ("poly", "mix"):
            assert abs(gp0 - kt.p0.grad.item()) < 1e-10 * max(1.0, abs(gp0))


@pytest.mark.parametrize("base", ["linear", "rbf", "poly", "matern32"])
@pytest.mark.parametrize("difference", [True, False])
@pytest.mark.parametrize("group,cols", [(16, 2), (16, 4), (64, 2)])
@pytest.mark.parametrize("scratch_free", [False, True, "lam"])
def test_wave_formulation_equals_autograd(base, difference, group, cols, scratch_free):
    """grad_wave_core.hpp: the two skewed sweeps with handed-over prefixes / suffixes, lock-step on the CPU; with the forward
    lattice kept (scratch), with the forward recursion undone on the way back and the gradient formed in the sweep
    (scratch-free), and with the forward recursion undone and Lam handed to the per-pair contraction ("lam")."""
    rng = np.random.default_rng(13)
    p0, p1 = _bp(base)
    cap = group * cols
    shapes = [(4, 2, 3, 6, 5, 3, "cross"), (3, 2, 2, 9, 9, 2, "sym"), (5, 3, 3, 7, 7, 5, "diag"), (1, 2, 2, 3, 4, 7, "cross"), (7, 2, 2, 6, 6, 2, "cross"),
              (3, 1, 2, 5, min(cap, 40) + (1 if difference else 0), 2, "cross"), (2, 2, 1, 30, 3, 3, "cross")]
    for (M, N1, N2, L1, L2, d, kind) in shapes:
        X = rng.standard_normal((N1, L1, d)) * 0.5
        Y = rng.standard_normal((N2, L2, d)) * 0.5 if kind == "cross" else None
        G = rng.standard_normal((M + 1, N1) if kind == "diag" else (M + 1, N1, N2 if kind == "cross" else N1))
        kt = _t_kern(base, d, M, difference=difference)
        tX = torch.tensor(X, requires_grad=True)
        tY = None if Y is None else torch.tensor(Y, requires_grad=True)
        lev = kt.K_seq_diag_levels(tX) if kind == "diag" else kt.K_seq_levels(tX, tY)
        (lev * torch.tensor(G)).sum().backward()
        if scratch_free is True and max(L1, L2) - (1 if difference else 0) > group * cols:
            continue          # both sides take the register role in turn
        gX, gY, gp0 = EG.seq_grad_wave(X, Y, G, M, base, difference, p0, p1, diag=(kind == "diag"), group=group, cols=cols, scratch_free=scratch_free)
        tol = 1e-9 if scratch_free else 1e-11
        assert rel(gX, tX.grad) < tol, (M, kind, rel(gX, tX.grad))
        if Y is not None:
            assert rel(gY, tY.grad) < tol
        if base == "poly":
            assert abs(gp0 - kt.p0.grad.item()) < 1e-10 * max(1.0, abs(gp0))


def test_torch_spectral_kernel_equals_numpy_oracle():
    """The torch restatement of SignatureSpectral's state-space kernel (gpsig/kernels.py:921-942), which the GPU gradient tests
    differentiate, against the NumPy oracle's: values on random and on coincident points, all three families; and its derivative at
    coincident points is finite (the reference's own, through tf.sqrt at 0, is not)."""
    import torch
    from oracle import sigkern_oracle as O, sigkern_oracle_torch as OT
    rng = np.random.default_rng(5)
    Q, d = 3, 4
    alpha, omega, gamma = np.exp(0.3 * rng.standard_normal(Q)), np.exp(0.3 * rng.standard_normal((Q, d))), np.exp(0.3 * rng.standard_normal((Q, d)))
    X, Y = rng.standard_normal((6, d)), rng.standard_normal((5, d))
    for family in ("rbf", "exp", "mixed"):
        for A, B in ((X, Y), (X, None)):
            want = O.base_spectral(A, B, alpha=alpha, omega=omega, gamma=gamma, family=family)
            tA = torch.tensor(A, requires_grad=True)
            got = OT.base_spectral(tA, None if B is None else torch.tensor(B), torch.tensor(alpha), torch.tensor(omega), torch.tensor(gamma), family)
            assert np.abs(got.detach().numpy() - want).max() < 1e-13
            got.sum().backward()
            assert bool(torch.isfinite(tA.grad).all())


@pytest.mark.parametrize("base", ["rbf", "linear", "matern32"])
@pytest.mark.parametrize("normalization,difference", [(True, True), (False, False)])
def test_torch_low_rank_restatement_equals_numpy_restatement(base, normalization, difference):
    """The differentiable low-rank restatement (landmarks gathered by index from the evaluation's scaled points) against the NumPy
    one (oracle/sigkern_oracle.py:LowRankOracle, given the landmark VALUES), same jitter draw and projections; and its gradient with
    respect to the inputs against central differences of itself (which moves the landmarks too, as tf.gather lets TensorFlow do)."""
    import types
    rng = np.random.default_rng(12)
    N, N2, L, d, M, T, c, r = 5, 3, 6, 2, 3, 4, 5, 4
    X, X2 = rng.standard_normal((N, L * d)) * 0.6, rng.standard_normal((N2, L * d)) * 0.6
    ls, var = rng.uniform(0.7, 1.5, d), rng.uniform(0.5, 1.5, M + 1)
    kn = _np_kern(base, d, M, normalization=normalization, difference=difference, lengthscales=ls, variances=var)
    kn.input_dim = L * d
    p0, p1 = _bp(base)
    kt = OT.LowRankTorchOracle(d, M, base, p0=torch.tensor(p0, dtype=torch.float64), p1=p1, normalization=normalization,
                               difference=difference, lengthscales=ls, variances=var)

    def sketch(k1, k2):
        nnz = 3 * r
        col = np.sort(rng.integers(0, r, nnz))
        colptr = np.concatenate(([0], np.cumsum(np.bincount(col, minlength=r))))
        return types.SimpleNamespace(k1=k1, k2=k2, r=r, colptr=colptr, i1=rng.integers(0, k1, nnz), i2=rng.integers(0, k2, nnz),
                                     val=rng.standard_normal(nnz))
    sks = [sketch(c, c)] + [sketch(c, r) for _ in range(M - 2)]
    jd = O.JITTER * rng.random(c)
    tol = 1e-9 if base != "linear" else 1e-5          # rank-deficient landmark Gram: INTEGRATION.md, "conventionally reproducible"
    Xs, X2s = kn._apply_scaling_and_lags_to_sequences(kn._seq3(X)), kn._apply_scaling_and_lags_to_sequences(kn._seq3(X2))
    # K(X), K(X, X2)
    idx = np.sort(rng.choice(N * L, c, replace=False))
    lo = O.LowRankOracle(kn, Xs.reshape(-1, d)[idx], jd, sks)
    assert rel(kt.set_draw(idx, jd, sks).K(torch.tensor(X)), lo.K(X)) < tol
    idx2 = np.sort(rng.choice((N + N2) * L, c, replace=False))
    lo = O.LowRankOracle(kn, np.concatenate([Xs.reshape(-1, d), X2s.reshape(-1, d)])[idx2], jd, sks)
    assert rel(kt.set_draw(idx2, jd, sks).K(torch.tensor(X), torch.tensor(X2), return_levels=True), lo.K(X, X2, return_levels=True)) < tol
    for incr in (False, True):
        Z = rng.standard_normal((M * (M + 1) // 2, T, 2, d) if incr else (M * (M + 1) // 2, T, d)) * 0.6
        Zs = kn._scale_Z(Z, incr)
        pool = np.concatenate([Zs.reshape(-1, d), Xs.reshape(-1, d)])
        idx3 = np.sort(rng.choice(pool.shape[0], c, replace=False))
        lo = O.LowRankOracle(kn, pool[idx3], jd, sks)
        kt.set_draw(idx3, jd, sks)
        assert rel(kt.K_tens_vs_seq(torch.tensor(Z), torch.tensor(X), increments=incr), lo.K_tens_vs_seq(Z, X, increments=incr)) < tol
        Kzz, Kzx, Kxx = kt.K_tens_n_seq_covs(torch.tensor(Z), torch.tensor(X), increments=incr)
        assert rel(Kzx, lo.K_tens_vs_seq(Z, X, increments=incr)) < tol and rel(Kzz, lo.K_tens(Z, increments=incr)) < tol
        assert rel(Kxx, lo.Kdiag(X)) < tol
    if base == "linear":
        return
    # gradient with respect to X through features AND landmarks: central differences of the restatement itself
    kt.set_draw(idx, jd, sks)
    Wm = torch.tensor(rng.standard_normal((N, N)))
    tX = torch.tensor(X, requires_grad=True)
    (kt.K(tX) * Wm).sum().backward()
    g = tX.grad.numpy()
    for (i, j) in [(0, 0), (2, 5), (4, L * d - 1), (idx[0] // L, (idx[0] % L) * d)]:       # the last one is a landmark's coordinate
        h = 1e-6
        Xp, Xm = X.copy(), X.copy()
        Xp[i, j] += h; Xm[i, j] -= h
        fd = (float((kt.K(torch.tensor(Xp)) * Wm).sum()) - float((kt.K(torch.tensor(Xm)) * Wm).sum())) / (2 * h)
        assert abs(fd - g[i, j]) < 1e-5 * max(1.0, np.abs(g).max()), (i, j, fd, g[i, j])


def test_lane_level_model_of_the_fused_reverse_kernel():
    """tools/sim_fused_grad.py replays grad_fused_kernel.hpp's interval schedule (evaluator / sweeper wavefronts, same-lane slots, the kernel-value
    ring, the skewed hand-overs) with arrays over lanes and checks both sides' gradients against torch.autograd of the plain recursion."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # lanes per pair group (four pairs per wavefront, or one pair with up to 256 points on the column side) x lattice columns per lane x
    # difference (1: the lattice of double increments, 0: the kernel matrix of the points)
    for lanes, cols, diff in (("16", "4", "1"), ("32", "4", "1"), ("64", "4", "1"), ("16", "2", "1"), ("16", "4", "0")):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "sim_fused_grad.py")], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, SIM_G=lanes, SIM_C=cols, SIM_DIFF=diff))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "worst" in r.stdout
