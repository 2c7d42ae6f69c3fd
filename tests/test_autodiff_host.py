"""Host-side pieces of gpsig_amd.autodiff that are plain torch (no GPU, no library call): the inducing tensors' rank-one level features against
the oracle's restatement of the reference notebook's construction (notebooks/signature_kernel.ipynb cell 18), and the custom autograd functions
of the level-feature route (per-level scaling, level norms, the product with the chunked backward) against torch.autograd.gradcheck."""
import numpy as np
import pytest
import torch

from gpsig_amd import autodiff
from oracle import sigkern_oracle as O


def test_rank_one_tensor_features_follow_the_notebook_construction():
    rng = np.random.default_rng(0)
    M, T, d = 4, 5, 3
    Z = rng.standard_normal((M * (M + 1) // 2, T, d))
    lev = autodiff._tensor_features(torch.tensor(Z), M, False, False)
    want = O.rank1_tensor_features(Z, M)                       # (T, 1 + d + .. + d^M), level 0 first
    got = np.concatenate([np.ones((T, 1))] + [a.numpy() for a in lev], axis=1)
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-14
    # increments: the difference of the component's two points (kernels.py:329-330 applied before the product, which is linear in it)
    Z2 = rng.standard_normal((M * (M + 1) // 2, T, 2, d))
    lev2 = autodiff._tensor_features(torch.tensor(Z2), M, True, False)
    want2 = O.rank1_tensor_features(Z2[:, :, 1] - Z2[:, :, 0], M)
    assert np.abs(np.concatenate([a.numpy() for a in lev2], axis=1) - want2[:, 1:]).max() < 1e-14
    # cosine: unit vectors first
    lev3 = autodiff._tensor_features(torch.tensor(Z), M, False, True)
    want3 = O.rank1_tensor_features(Z / np.linalg.norm(Z, axis=-1, keepdims=True), M)
    assert np.abs(np.concatenate([a.numpy() for a in lev3], axis=1) - want3[:, 1:]).max() < 1e-14


def test_column_levels_of_a_feature_buffer():
    col = autodiff._col_levels(3, 3, 48, torch.device("cpu")).numpy()
    assert list(col[:3]) == [1] * 3 and list(col[3:12]) == [2] * 9 and list(col[12:39]) == [3] * 27 and col[39] == 0 and set(col[40:]) == {4}


def _buffer(rng, N, d, M):
    F = sum(d ** m for m in range(1, M + 1))
    ld = (F + 1 + 15) // 16 * 16
    Phi = np.zeros((N, ld))
    Phi[:, :F] = rng.standard_normal((N, F))
    Phi[:, F] = 1.0
    return torch.tensor(Phi, requires_grad=True), F, ld


def test_level_scaling_and_norms_pass_gradcheck():
    rng = np.random.default_rng(1)
    N, d, M = 4, 2, 3
    Phi, F, ld = _buffer(rng, N, d, M)
    fac = torch.tensor(rng.uniform(0.5, 1.5, (M + 1, N)), requires_grad=True)
    out = autodiff._ScaleLevels.apply(Phi, fac, d)
    off = 0
    for m in range(1, M + 1):
        assert torch.allclose(out[:, off:off + d ** m], Phi[:, off:off + d ** m] * fac[m][:, None])
        off += d ** m
    assert torch.allclose(out[:, F], fac[0]) and float(out.detach()[:, F + 1:].abs().max()) == 0.0
    assert torch.autograd.gradcheck(lambda P, f: autodiff._ScaleLevels.apply(P, f, d), (Phi, fac), atol=1e-8)
    nm = autodiff._LevelNorms.apply(Phi, d, M)
    assert nm.shape == (M + 1, N) and torch.allclose(nm[0], torch.ones(N, dtype=torch.float64))
    assert torch.allclose(nm[2], (Phi[:, d:d + d * d] ** 2).sum(dim=1))
    # level 0 is the constant 1 of the buffer: no gradient flows into its column (gradcheck would see 2 there: compared by hand)
    W = torch.tensor(rng.standard_normal((M + 1, N)))
    (g,) = torch.autograd.grad((nm * W).sum(), Phi)
    want = torch.zeros_like(Phi)
    off = 0
    for m in range(1, M + 1):
        want[:, off:off + d ** m] = 2.0 * Phi.detach()[:, off:off + d ** m] * W[m][:, None]
        off += d ** m
    assert torch.allclose(g, want)


@pytest.mark.parametrize("N", [7, 512, 1024])
def test_feature_product_backward_in_chunks(N):
    """_FeatureProduct: dA as a batch of products over chunks of the long axis (N a multiple of 2 .. 32 with at least 256 per chunk) or as one
    product; a stride-0 upstream (what .sum() hands back) first made contiguous."""
    rng = np.random.default_rng(2)
    A = torch.tensor(rng.standard_normal((5, 16)), requires_grad=True)
    B = torch.tensor(rng.standard_normal((N, 16)), requires_grad=True)
    out = autodiff._FeatureProduct.apply(A, B)
    assert torch.allclose(out, A @ B.T)
    out.sum().backward()                                                # expanded upstream
    ones = torch.ones(5, N, dtype=torch.float64)
    assert torch.allclose(A.grad, ones @ B.detach()) and torch.allclose(B.grad, ones.T @ A.detach())
    W = torch.tensor(rng.standard_normal((5, N)))
    A.grad = B.grad = None
    (autodiff._FeatureProduct.apply(A, B) * W).sum().backward()
    assert torch.allclose(A.grad, W @ B.detach()) and torch.allclose(B.grad, W.T @ A.detach())


def test_host_copy_of_the_level_weights_is_dropped_by_load_state_dict():
    """Round 6: the weights' host memo is keyed on the parameters' version counters, which load_state_dict() and ``.data`` writes
    do not bump; the module drops the memo on load_state_dict() and offers invalidate_host_copies() for the rest."""
    from gpsig_amd import kernels
    mod = autodiff.SignatureKernelModule(kernels.SignatureLinear(12, 3, 3), device="cpu")
    mod._w_host_memo = ("key", "stale")
    mod.load_state_dict(mod.state_dict())
    assert mod._w_host_memo is None
    mod._w_host_memo = ("key", "stale")
    mod.invalidate_host_copies()
    assert mod._w_host_memo is None
