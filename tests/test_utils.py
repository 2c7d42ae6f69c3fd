"""Host-side initialisers (gpsig_amd/utils.py): layouts the kernels consume and the lengthscale heuristic."""
import numpy as np

from gpsig_amd import utils


def test_inducing_tensor_layout_and_class_shares():
    rng = np.random.default_rng(0)
    N, L, d, M = 30, 12, 3, 4
    X = rng.standard_normal((N, L, d))
    y = np.repeat([0, 1, 2], 10)
    lt = M * (M + 1) // 2
    assert utils.suggest_initial_inducing_tensors(X, M, 17, rng=1).shape == (lt, 17, d)
    assert utils.suggest_initial_inducing_tensors(X, M, 17, labels=y, increments=True, rng=1).shape == (lt, 17, 2, d)
    assert utils.suggest_initial_inducing_tensors(X, M, 8, increments=True, num_lags=2, rng=1).shape == (lt, 8, 2, 3 * d)
    assert utils.suggest_initial_inducing_tensors(X, M, 8, num_lags=1, rng=1).shape == (lt, 8, 2 * d)
    # without the jitter every component is an observation of the data, in time order within a level
    Xi = np.arange(N * L, dtype=np.float64).reshape(N, L, 1) * np.ones((1, 1, d))

    class NoNoise(np.random.Generator):
        def standard_normal(self, size=None, *a, **k):
            return np.zeros(size)
    Z = utils.suggest_initial_inducing_tensors(Xi, M, 6, increments=True, rng=NoNoise(np.random.PCG64(3)))
    assert np.all(Z[:, :, 1, 0] - Z[:, :, 0, 0] == 1)                    # second point = next observation
    k = 0
    for m in range(1, M + 1):
        blk = Z[k:k + m, :, 0, 0]
        assert np.all(np.diff(blk, axis=0) > 0)                            # sorted in time within the level
        assert np.all(blk // L == blk[0] // L)                             # all from one sequence
        k += m


def test_inducing_sequences_and_lengthscales():
    rng = np.random.default_rng(1)
    X = rng.standard_normal((20, 15, 2))
    X[3, 9:] = np.nan
    Z = utils.suggest_initial_inducing_sequences(X, 11, 5, labels=np.repeat([0, 1], 10), rng=2)
    assert Z.shape == (11, 5, 2) and not np.any(np.isnan(Z))
    P = X.reshape(-1, 2)
    P = P[~np.any(np.isnan(P), axis=1)]
    want = np.sqrt(np.mean((P[:, None, :] - P[None, :, :]) ** 2, axis=(0, 1)) * 2)       # utils.py:93-94 literally
    assert np.allclose(utils.suggest_initial_lengthscales(X), want, rtol=1e-12)
