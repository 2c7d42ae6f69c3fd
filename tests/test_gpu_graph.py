"""HIP-graph capture of evaluations through the C ABI (include/gpsig_hip.h: gpsig_graph_begin / _end / _launch)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from gpsig_amd import kernels
    return torch, kernels, torch.device("cuda:0")


def test_replay_equals_eager_evaluation(env):
    torch, K, dev = env
    rng = np.random.default_rng(5)
    N, N2, L, d, M, T = 24, 9, 20, 3, 4, 7
    for cls, kw in ((K.SignatureRBF, dict(lengthscales=[0.8, 1.1, 1.3])), (K.SignatureLinear, dict(num_lags=1)), (K.SignatureMatern32, dict(order=2))):
        kern = cls(L * d, d, M, **kw)
        X = torch.tensor(rng.standard_normal((N, L * d)) * 0.4, device=dev)
        X2 = torch.tensor(rng.standard_normal((N2, L * d)) * 0.4, device=dev)
        Z = torch.tensor(rng.standard_normal((M * (M + 1) // 2, T, d * ((kern.num_lags or 0) + 1))) * 0.4, device=dev)
        calls = [("K", (X,)), ("K", (X, X2)), ("Kdiag", (X,)), ("K_tens", (Z,)), ("K_tens_vs_seq", (Z, X))]
        graphs = [kern.graphed(m, *a) for m, a in calls]
        for (m, a), g in zip(calls, graphs):                     # recorded on the first contents
            assert torch.equal(g.out, getattr(kern, m)(*a)), m
        Xn = torch.tensor(rng.standard_normal((N, L * d)) * 0.4, device=dev)
        Zn = torch.tensor(rng.standard_normal(tuple(Z.shape)) * 0.4, device=dev)
        X.copy_(Xn); Z.copy_(Zn)
        for (m, a), g in zip(calls, graphs):                     # replayed on new contents of the same buffers
            got = g.replay().clone()
            assert torch.equal(got, getattr(kern, m)(*a)), m
            assert not torch.equal(got, torch.zeros_like(got))


def test_capture_refuses_what_needs_the_host(env):
    torch, K, dev = env
    from gpsig_amd import _lib
    rng = np.random.default_rng(6)
    L, d, M = 12, 2, 3
    kern = K.SignatureRBF(L * d, d, M)
    X = torch.tensor(rng.standard_normal((40, L * d)), device=dev)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        ctx = _lib.context(0, s.cuda_stream)
        ctx.set_pointer_mode(_lib.PTR_DEVICE)
        # nothing was evaluated on this context yet: buffers and task lists are missing
        with pytest.raises(ValueError, match="graph capture"):
            with ctx.graph():
                kern.K(X)
        want = kern.K(X)                                          # the context works as before
        with ctx.graph() as g:
            out = kern.K(X)
        out.zero_()
        g.launch()
        assert torch.equal(out, want)
        # changed hyper-parameters need an upload
        kern.variances = np.asarray(kern.variances) * 2.0
        with pytest.raises(ValueError, match="level weights changed"):
            with ctx.graph():
                kern.K(X)
        assert torch.allclose(kern.K(X), 2.0 * want, rtol=1e-12)
        kern.variances = np.asarray(kern.variances) / 2.0
        kern.K(X)
        # a larger evaluation moves scratch buffers: the old graph must not be replayed
        big = torch.tensor(rng.standard_normal((400, 4 * L * d)), device=dev)
        K.SignatureRBF(4 * L * d, d, M).K(big)
        with pytest.raises(ValueError, match="scratch buffers moved"):
            g.launch()
        s.synchronize()
    # host-pointer contexts cannot record
    hctx = _lib.context(0, 0)
    hctx.set_pointer_mode(_lib.PTR_HOST)
    with pytest.raises(ValueError, match="device-pointer mode"):
        with hctx.graph():
            pass
    dctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
    dctx.set_pointer_mode(_lib.PTR_DEVICE)
    if torch.cuda.current_stream(dev).cuda_stream == 0:
        with pytest.raises(ValueError, match="default stream"):
            with dctx.graph():
                pass


def test_recordings_release_their_context(env):
    torch, K, dev = env
    from gpsig_amd import _lib
    kern = K.SignatureRBF(12 * 2, 2, 3)
    X = torch.tensor(np.random.default_rng(7).standard_normal((16, 24)), device=dev)
    before = len(_lib._contexts)
    for _ in range(5):
        g = kern.graphed("K", X)
        want = kern.K(X)
        assert torch.equal(g.replay(), want)
        del g
    import gc
    gc.collect()
    assert len(_lib._contexts) <= before + 1


def test_replay_of_the_feature_contraction(env):
    """SignatureLinear's Gram through the matrix-core contraction of explicit level features (three kernels, no host step once its
    scratch exists) records and replays like the pair recursion."""
    torch, K, dev = env
    rng = np.random.default_rng(6)
    N, N2, L, d, M = 400, 300, 64, 8, 3               # 80,200 / 120,000 pairs of 64-point sequences: the planner's feature route
    kern = K.SignatureLinear(L * d, d, M, lengthscales=0.8 + 0.1 * np.arange(d))
    X = torch.tensor(rng.standard_normal((N, L * d)) * 0.4, device=dev)
    X2 = torch.tensor(rng.standard_normal((N2, L * d)) * 0.4, device=dev)
    g1, g2 = kern.graphed("K", X), kern.graphed("K", X, X2)
    from gpsig_amd import _lib
    ctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
    try:                                                 # what ran is the contraction: the recursion gives other last digits
        ctx.set_option("sig_features", 0)
        lattice = kern.K(X)
    finally:
        ctx.set_option("sig_features", -1)
    assert not torch.equal(g1.out, lattice) and float((g1.out - lattice).abs().max()) < 1e-12 * float(lattice.abs().max())
    X.copy_(torch.tensor(rng.standard_normal((N, L * d)) * 0.4, device=dev))
    a, b = g1.replay().clone(), g2.replay().clone()
    assert torch.equal(a, kern.K(X)) and torch.equal(b, kern.K(X, X2))


def test_float32_requests_computed_in_float64_are_refused_by_capture(env):
    """Round 6: float32 evaluations of SignatureCosine and of one-column state spaces are computed in float64 and rounded
    (kernels._f32_upcast); their conversions would be torch temporaries recorded inside the capture.  graphed() refuses them; float32
    requests the float32 kernels take as they are, and the float64 form of the refused ones, record and replay."""
    torch, K, dev = env
    rng = np.random.default_rng(8)
    N, L, M = 12, 16, 3
    for kern, d in ((K.SignatureCosine(L * 2, 2, M), 2), (K.SignatureRBF(L * 1, 1, M), 1)):
        X32 = torch.tensor(rng.standard_normal((N, L * d)) * 0.5, device=dev, dtype=torch.float32)
        with pytest.raises(ValueError, match="computed in float64"):
            kern.graphed("K", X32)
        assert not kern._graph_recording
        eager = kern.K(X32)                                           # the eager float32 request still works (upcast + rounding)
        assert eager.dtype == torch.float32
        X64 = X32.double()
        g = kern.graphed("K", X64)
        assert torch.equal(g.replay(), kern.K(X64))
        assert float((g.out.float() - eager).abs().max()) < 1e-5
    kern = K.SignatureRBF(L * 3, 3, M)
    X32 = torch.tensor(rng.standard_normal((N, L * 3)) * 0.5, device=dev, dtype=torch.float32)
    g = kern.graphed("K", X32)                                        # float32 kernels proper: recorded as they are
    X32.copy_(torch.tensor(rng.standard_normal((N, L * 3)) * 0.5, device=dev, dtype=torch.float32))
    assert torch.equal(g.replay().clone(), kern.K(X32))
