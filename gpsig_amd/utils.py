"""Initial values for the inducing variables and lengthscales (host side, NumPy): the call surface of ``gpsig.utils``.

Reference: gpsig/utils.py:6-97.  These heuristics only matter here because they define the LAYOUT the kernels consume:
inducing tensors are (M(M+1)/2, num_inducing, d') -- level m contributes m consecutive components, each an observation
drawn (in time order) from a training sequence -- or (M(M+1)/2, num_inducing, 2, d') with increments, the second point
being the next observation (utils.py:9-22, :43).  Randomness comes from ``rng`` (a ``numpy.random.Generator``; the reference
uses NumPy's global state), so the draws are reproducible but not the reference's.
"""
import numpy as np


def _rng(rng):
    return rng if isinstance(rng, np.random.Generator) else np.random.default_rng(rng)


def _class_quota(labels, n_total, num_inducing):
    """utils.py:30-36 / :73-79: floor(share of the class * num_inducing) per class, in label order."""
    counts = np.bincount(np.asarray(labels))
    return [(c, int(np.floor(float(n_c) / n_total * num_inducing))) for c, n_c in enumerate(counts)]


def _draw_tensors(seqs, count, num_levels, increments, rng):
    """One block (count, M(M+1)/2, [2,] d): for level m, m observation indices without replacement, sorted in time."""
    picked = seqs[rng.integers(0, seqs.shape[0], size=count)]                     # utils.py:8 (with replacement)
    L = picked.shape[1]
    blocks = []
    for m in range(1, num_levels + 1):
        idx = np.stack([np.sort(rng.choice(L - 1 if increments else L, size=m, replace=False)) for _ in range(count)], axis=0) \
            if count else np.zeros((0, m), dtype=np.int64)                          # utils.py:11-12 / :18-19
        first = np.take_along_axis(picked, idx[:, :, None], axis=1)
        if increments:
            second = np.take_along_axis(picked, idx[:, :, None] + 1, axis=1)        # utils.py:14
            blocks.append(np.stack((first, second), axis=2))
        else:
            blocks.append(first)
    return np.concatenate(blocks, axis=1)


def suggest_initial_inducing_tensors(sequences, num_levels, num_inducing, labels=None, increments=False, num_lags=None, rng=None):
    """Reference: utils.py:25-52.  sequences (N, L, d).  Returns Z of shape (M(M+1)/2, num_inducing, [2,] d * (num_lags + 1)),
    observations of the data jittered by 0.4 * N(0, 1); with labels every class gets its share of the inducing tensors."""
    rng = _rng(rng)
    sequences = np.asarray(sequences, dtype=np.float64)
    parts = []
    if labels is not None:
        labels = np.asarray(labels)
        for c, quota in _class_quota(labels, sequences.shape[0], num_inducing):
            parts.append(_draw_tensors(sequences[labels == c], quota, num_levels, increments, rng))
    missing = num_inducing - sum(p.shape[0] for p in parts)
    if missing > 0:
        parts.append(_draw_tensors(sequences, missing, num_levels, increments, rng))
    Z = np.concatenate(parts, axis=0)                                               # (num_inducing, lt, [2,] d)
    Z = np.moveaxis(Z, 0, 1)                                                        # utils.py:43: components first
    if num_lags is not None and num_lags > 0:                                       # utils.py:44-48: one copy per lag
        Z = np.repeat(Z[..., None, :], num_lags + 1, axis=-2).reshape(*Z.shape[:-1], -1)
    return Z + 0.4 * rng.standard_normal(Z.shape)                                   # utils.py:50


def _draw_sequences(seqs, count, len_inducing, rng):
    """utils.py:54-62: windows of len_inducing consecutive observations ending before the NaN padding starts."""
    picked = seqs[rng.integers(0, seqs.shape[0], size=count)]
    nan_row = np.any(np.isnan(picked), axis=2)
    first_nan = np.where(nan_row.any(axis=1), np.argmax(nan_row, axis=1), seqs.shape[1])
    last = np.array([rng.integers(len_inducing - 1, first_nan[i]) for i in range(count)], dtype=np.int64).reshape(count)
    idx = last[:, None] - len_inducing + 1 + np.arange(len_inducing)[None, :]
    return np.take_along_axis(picked, idx[:, :, None], axis=1)


def suggest_initial_inducing_sequences(sequences, num_inducing, len_inducing, labels=None, rng=None):
    """Reference: utils.py:65-86.  Returns Z (num_inducing, len_inducing, d)."""
    rng = _rng(rng)
    sequences = np.asarray(sequences, dtype=np.float64)
    parts = []
    if labels is not None:
        labels = np.asarray(labels)
        for c, quota in _class_quota(labels, sequences.shape[0], num_inducing):
            parts.append(_draw_sequences(sequences[labels == c], quota, len_inducing, rng))
    missing = num_inducing - sum(p.shape[0] for p in parts)
    if missing > 0:
        parts.append(_draw_sequences(sequences, missing, len_inducing, rng))
    Z = np.concatenate(parts, axis=0)
    return Z + 0.4 * rng.standard_normal(Z.shape)


def suggest_initial_lengthscales(X, num_samples=None, rng=None):
    """Reference: utils.py:88-97: per feature, sqrt(d * mean squared pairwise difference of the observations).
    Computed from first and second moments instead of the (P, P, d) difference tensor: mean_ij (x_i - x_j)^2 = 2 (E x^2 - (E x)^2)."""
    rng = _rng(rng)
    X = np.asarray(X, dtype=np.float64).reshape(-1, np.asarray(X).shape[-1])
    X = X[~np.any(np.isnan(X), axis=1)]
    if num_samples is not None and num_samples < X.shape[0]:
        X = X[rng.choice(X.shape[0], size=num_samples, replace=False)]
    msd = 2.0 * (np.mean(X * X, axis=0) - np.mean(X, axis=0) ** 2)
    return np.sqrt(msd * X.shape[1])
