"""CPU oracle for the GPSig signature-kernel evaluation path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  ``gpsig_amd`` never does: the product path is the HIP library and
it fails loudly when that library is missing.

What it is: a NumPy (fp64 by default) restatement, operation for operation, of
the TensorFlow graph the reference builds for the hot path.  Every function
names the reference lines it follows (paths relative to ``/root/reference``).
The arithmetic of the reference lives in TensorFlow 1.15.3 / GPflow 1.5.1
(``requirements.txt:7-8``), which are not vendored and cannot be installed in
this image; the TF ops involved (matmul, strided slice, exclusive cumsum,
reduce_sum, exp, sqrt, divide) are elementary and deterministic up to
summation order, so a NumPy transcription is semantically exact.

Parity pinning: the reference ships NO golden vectors (its only check,
``notebooks/signature_kernel.ipynb``, uses unseeded data and prints matrix
norms).  This oracle is therefore pinned BY PROPERTY, by the same three
identities the notebook checks against esig (cells 6-29), using the
independent routines at the bottom of this file (`truncated_signature`,
`rank1_tensor_features`, `brute_force_first_order`).  Everything the notebook
does not exercise (order=1 default, non-linear base kernels, normalisation,
X2 != None, increments, lags) is "parity unpinned" in the reference itself and
is pinned only by this restatement of the cited lines.

Deliberate deviations (reference bugs that are NOT reproduced) are
marked ``# Q6``.
"""
from __future__ import annotations

import numpy as np

JITTER = 1e-6  # gpflow.settings.jitter == settings.numerics.jitter_level (GPflow 1.5.1 default)


# ---------------------------------------------------------------------------
# small TF-op equivalents
# ---------------------------------------------------------------------------
def _excumsum(A, axis):
    """tf.cumsum(A, exclusive=True, axis=axis)."""
    if A.shape[axis] == 0:
        return A.copy()
    out = np.cumsum(A, axis=axis)
    out = np.roll(out, 1, axis=axis)
    idx = [slice(None)] * A.ndim
    idx[axis] = 0
    out[tuple(idx)] = 0
    return out


def _double_increment(M):
    """M[:, 1:, ..., 1:] + M[:, :-1, ..., :-1] - M[:, :-1, ..., 1:] - M[:, 1:, ..., :-1]
    (gpsig/signature_algs.py:26 and :56)."""
    return M[:, 1:, ..., 1:] + M[:, :-1, ..., :-1] - M[:, :-1, ..., 1:] - M[:, 1:, ..., :-1]


# ---------------------------------------------------------------------------
# gpsig/signature_algs.py
# ---------------------------------------------------------------------------
def signature_kern_first_order(M, num_levels, difference=True):
    """gpsig/signature_algs.py:8-35.

    M: (N1, L1, N2, L2) or (N, L, L).  Returns (num_levels+1, N1, N2) or
    (num_levels+1, N).
    """
    M = np.asarray(M)
    if M.ndim == 4:
        K = [np.ones((M.shape[0], M.shape[2]), dtype=M.dtype)]          # :19-20
    else:
        K = [np.ones((M.shape[0],), dtype=M.dtype)]                     # :22-23
    if difference:
        M = _double_increment(M)                                        # :26
    K.append(M.sum(axis=(1, -1)))                                       # :28
    R = M                                                               # :30
    for _ in range(2, num_levels + 1):                                  # :31
        R = M * _excumsum(_excumsum(R, 1), -1)                          # :32
        K.append(R.sum(axis=(1, -1)))                                   # :33
    return np.stack(K, axis=0)                                          # :35


def signature_kern_higher_order(M, num_levels, order=2, difference=True):
    """gpsig/signature_algs.py:37-74."""
    M = np.asarray(M)
    if M.ndim == 4:
        K = [np.ones((M.shape[0], M.shape[2]), dtype=M.dtype)]          # :49-50
    else:
        K = [np.ones((M.shape[0],), dtype=M.dtype)]                     # :52-53
    if difference:
        M = _double_increment(M)                                        # :56
    K.append(M.sum(axis=(1, -1)))                                       # :58
    R = [[M]]                                                           # :60  (1 x 1 grid)
    for i in range(2, num_levels + 1):                                  # :61
        d = min(i, order)                                               # :62
        rows_prev, cols_prev = len(R), len(R[0])
        Rn = [[None] * d for _ in range(d)]                             # :63
        total = sum(R[r][c] for r in range(rows_prev) for c in range(cols_prev))
        Rn[0][0] = M * _excumsum(_excumsum(total, 1), -1)               # :64
        for j in range(2, d + 1):                                       # :65
            col_sum = sum(R[r][j - 2] for r in range(rows_prev))        # R[:, j-2]
            row_sum = sum(R[j - 2][c] for c in range(cols_prev))        # R[j-2, :]
            Rn[0][j - 1] = 1.0 / j * M * _excumsum(col_sum, 1)          # :66
            Rn[j - 1][0] = 1.0 / j * M * _excumsum(row_sum, -1)         # :67
            for k in range(2, d + 1):                                   # :68
                Rn[j - 1][k - 1] = 1.0 / (j * k) * M * R[j - 2][k - 2]  # :69
        K.append(sum(Rn[r][c] for r in range(d) for c in range(d)).sum(axis=(1, -1)))  # :71
        R = Rn                                                          # :72
    return np.stack(K, axis=0)                                          # :74


def tensor_kern(M, num_levels):
    """gpsig/signature_algs.py:76-99.  M: (lt, T, T')."""
    M = np.asarray(M)
    K = [np.ones(M.shape[1:], dtype=M.dtype)]                           # :88
    k = 0
    for i in range(1, num_levels + 1):                                  # :91
        R = M[k]; k += 1                                                # :92-93
        for _ in range(1, i):                                           # :94
            R = M[k] * R; k += 1                                        # :95-96
        K.append(R)                                                     # :97
    return np.stack(K, axis=0)


def signature_kern_tens_vs_seq_first_order(M, num_levels, difference=True):
    """gpsig/signature_algs.py:101-127.  M: (lt, T, N, L)."""
    M = np.asarray(M)
    if difference:
        M = M[..., 1:] - M[..., :-1]                                    # :114
    K = [np.ones(M.shape[1:3], dtype=M.dtype)]                          # :116
    k = 0
    for i in range(1, num_levels + 1):                                  # :119
        R = M[k]; k += 1                                                # :120-121
        for _ in range(1, i):                                           # :122
            R = M[k] * _excumsum(R, 2); k += 1                          # :123-124
        K.append(R.sum(axis=2))                                         # :125
    return np.stack(K, axis=0)


def signature_kern_tens_vs_seq_higher_order(M, num_levels, order=2, difference=True):
    """gpsig/signature_algs.py:129-160."""
    M = np.asarray(M)
    if difference:
        M = M[..., 1:] - M[..., :-1]                                    # :142
    K = [np.ones(M.shape[1:3], dtype=M.dtype)]                          # :144
    k = 0
    for i in range(1, num_levels + 1):                                  # :147
        R = [M[k]]; k += 1                                              # :148-149
        for j in range(1, i):                                           # :150
            d = min(j + 1, order)                                       # :151
            Rn = [None] * d
            Rn[0] = M[k] * _excumsum(sum(R), 2)                         # :153
            for l in range(1, d):                                       # :154
                Rn[l] = 1.0 / (l + 1) * M[k] * R[l - 1]                 # :155
            R = Rn; k += 1                                              # :156-157
        K.append(sum(R).sum(axis=2))                                    # :158
    return np.stack(K, axis=0)


# ---------------------------------------------------------------------------
# gpsig/lags.py
# ---------------------------------------------------------------------------
def lin_interp(time, X, time_query):
    """gpsig/lags.py:7-38 (3-D X branch :32-33).  X: (N, L, d); time: (L,);
    time_query: (L, p).  Returns (N, L, p, d)."""
    dist = time[:, None, None] - time_query[None, :, :]                               # :20
    masked = np.where(dist > JITTER, -np.inf, dist)                                   # :22
    left = np.argmax(masked, axis=0)                                                  # :22  (L, p)
    right = left + 1                                                                  # :23
    Xl, Xr = X[:, left, :], X[:, np.minimum(right, X.shape[1] - 1), :]                # :25-26
    tl, tr = time[left], time[np.minimum(right, time.shape[0] - 1)]                   # :28-29
    # NOTE tf.gather on CPU raises for right_idx == L; that only happens when a query equals
    # the last time point, which add_lags_to_sequences never produces for lags > 0.
    return Xl + (time_query[None, ..., None] - tl[None, ..., None]) * (Xr - Xl) / (tr[None, ..., None] - tl[None, ..., None])  # :33


def add_lags_to_sequences(X, lags):
    """gpsig/lags.py:41-63.  X: (N, L, d) -> (N, L, p+1, d)."""
    L = X.shape[1]
    time = np.arange(L, dtype=X.dtype) / (L - 1)                                      # :56
    time_lags = np.maximum(time[:, None] - np.asarray(lags, dtype=X.dtype)[None, :], 0.)  # :57
    X_lags = lin_interp(time, X, time_lags)                                           # :59
    return np.concatenate((X[:, :, None, :], X_lags), axis=2)                         # :61


# ---------------------------------------------------------------------------
# gpsig/kernels.py  -- base kernels (static kernels on R^d)
# ---------------------------------------------------------------------------
def _mm_t(X, X2):
    """tf.matmul(X, X2, transpose_b=True), batched over leading axes."""
    return np.matmul(X, np.swapaxes(X2, -1, -2))


def _square_dist(X, X2=None):
    """gpsig/kernels.py:765-776."""
    Xs = np.sum(np.square(X), axis=-1)
    if X2 is None:
        dist = -2 * _mm_t(X, X)
        dist += Xs[..., :, None] + Xs[..., None, :]
        return dist
    X2s = np.sum(np.square(X2), axis=-1)
    dist = -2 * _mm_t(X, X2)
    dist += Xs[..., :, None] + X2s[..., None, :]
    return dist


def _euclid_dist(X, X2=None):
    """gpsig/kernels.py:779-781."""
    return np.sqrt(np.maximum(_square_dist(X, X2), 1e-40))


def base_lin(X, X2=None, **_):
    """gpsig/kernels.py:799-806."""
    return _mm_t(X, X if X2 is None else X2)


def base_cos(X, X2=None, **_):
    """gpsig/kernels.py:820-828."""
    Xn = np.sqrt(np.sum(np.square(X), axis=-1))
    if X2 is None:
        return _mm_t(X, X) / (Xn[..., :, None] * Xn[..., None, :])
    X2n = np.sqrt(np.sum(np.square(X2), axis=-1))
    return _mm_t(X, X2) / (Xn[..., :, None] * X2n[..., None, :])


def base_poly(X, X2=None, gamma=1.0, degree=3.0, **_):
    """gpsig/kernels.py:844-848."""
    return (_mm_t(X, X if X2 is None else X2) + gamma) ** degree


def base_rbf(X, X2=None, **_):
    """gpsig/kernels.py:862-864."""
    return np.exp(-_square_dist(X, X2) / 2)


def base_mix(X, X2=None, mixing=0.5, **_):
    """gpsig/kernels.py:881-892."""
    Xs = np.sum(np.square(X), axis=-1)
    if X2 is None:
        inner = _mm_t(X, X)
        ds = Xs[..., :, None] + Xs[..., None, :] - 2 * inner
    else:
        X2s = np.sum(np.square(X2), axis=-1)
        inner = _mm_t(X, X2)
        ds = Xs[..., :, None] + X2s[..., None, :] - 2 * inner
    return mixing * np.exp(-ds / 2) + (1. - mixing) * inner


def base_matern12(X, X2=None, **_):
    """gpsig/kernels.py:955-958."""
    return np.exp(-_euclid_dist(X, X2))


def base_matern32(X, X2=None, **_):
    """gpsig/kernels.py:974-977."""
    r = _euclid_dist(X, X2)
    return (1. + np.sqrt(3.) * r) * np.exp(-np.sqrt(3.) * r)


def base_matern52(X, X2=None, **_):
    """gpsig/kernels.py:991-993."""
    r = _euclid_dist(X, X2)
    return (1.0 + np.sqrt(5.) * r + 5. / 3. * np.square(r)) * np.exp(-np.sqrt(5.) * r)


def base_spectral(X, X2=None, alpha=None, omega=None, gamma=None, family="rbf", **_):
    """gpsig/kernels.py:921-942.  X (P, d), X2 (P2, d) or batched (N, L, d) [the reference's tile() works on 2-D inputs only;
    batching is the evident extension].  'mixed': Q6 -- the reference uses an undefined Q and flips a sign (:932-936); the
    intent (first floor(Q/2) components Gaussian, the rest exponential) is implemented."""
    X2 = X if X2 is None else X2
    diff = X[..., :, None, :] - X2[..., None, :, :]                                                   # (..., P, P2, d)
    Q = alpha.shape[0]
    out = 0.0
    for q in range(Q):
        sq = np.sum(np.square(diff * gamma[q]), axis=-1)
        if family == "rbf" or (family == "mixed" and q < Q // 2):
            env = np.exp(-sq / 2)                                                                     # :926
        else:
            env = np.exp(-np.sqrt(sq) / 2)                                                            # :924
        out = out + alpha[q] * env * np.cos(2. * np.pi * np.sum(diff * omega[q], axis=-1))            # :937, :942
    return out


BASE_KERNELS = {
    "linear": base_lin, "cosine": base_cos, "poly": base_poly, "rbf": base_rbf, "mix": base_mix,
    "matern12": base_matern12, "matern32": base_matern32, "matern52": base_matern52, "spectral": base_spectral,
}


# ---------------------------------------------------------------------------
# gpsig/kernels.py  -- SignatureKernel
# ---------------------------------------------------------------------------
class SignatureKernelOracle:
    """NumPy restatement of ``gpsig.kernels.SignatureKernel`` (kernels.py:15-761) with the exact
    (non low-rank) branches.  Hyper-parameters are plain constrained values."""

    def __init__(self, input_dim, num_features, num_levels, base="linear", variances=1, lengthscales=1,
                 order=1, normalization=True, difference=True, num_lags=None, base_params=None,
                 dtype=np.float64):
        if input_dim % num_features != 0:                                                  # :98-101
            raise ValueError("The arguments num_features and input_dim are not consistent.")
        self.input_dim, self.num_features, self.num_levels = input_dim, num_features, num_levels
        self.len_examples = input_dim // num_features
        self.order = num_levels if (order <= 0 or order >= num_levels) else order           # :57
        self.normalization, self.difference = normalization, difference
        self.dtype = dtype
        self.variances = np.asarray(variances * np.ones(num_levels + 1), dtype=np.float64)  # :65,:129
        self.sigma = 1.0                                                                   # :66
        if num_lags is None:
            self.num_lags = 0
        else:
            if not isinstance(num_lags, int) or num_lags < 0:                              # :74-75
                raise ValueError('The variable num_lags most be a nonnegative integer or None.')
            self.num_lags = num_lags
            if num_lags > 0:
                self.lags = 0.1 * np.arange(1, num_lags + 1, dtype=np.float64)             # :79
                g = 1. / np.arange(1, num_lags + 2, dtype=np.float64)                      # :80
                self.gamma = g / g.sum()                                                   # :81
        self.lengthscales = None if lengthscales is None else np.asarray(
            lengthscales * np.ones(num_features), dtype=np.float64)                        # :84-88
        self.base = base
        self.base_params = dict(base_params or {})

    # -- base kernel dispatch --------------------------------------------------------------
    def _base_kern(self, X, X2=None):
        return BASE_KERNELS[self.base](X, X2, **self.base_params)

    # -- scaling (kernels.py:343-398) ------------------------------------------------------
    def _apply_scaling_and_lags_to_sequences(self, X):
        N, L, _ = X.shape
        nf = self.num_features * (self.num_lags + 1)                                       # :350
        if self.num_lags > 0:
            X = add_lags_to_sequences(X, self.lags)                                        # :353
        X = X.reshape(N, L, self.num_lags + 1, self.num_features)                          # :355
        if self.lengthscales is not None:
            X = X / self.lengthscales[None, None, None, :].astype(X.dtype)                 # :358
        if self.num_lags > 0:
            X = X * self.gamma[None, None, :, None].astype(X.dtype)                        # :361
        return X.reshape(N, L, nf)                                                         # :363

    def _apply_scaling_to_tensors(self, Z):
        lt, T = Z.shape[0], Z.shape[1]
        if self.lengthscales is not None:                                                  # :374
            Z = Z.reshape(lt, T, self.num_lags + 1, self.num_features)
            Z = Z / self.lengthscales[None, None, None, :].astype(Z.dtype)
            if self.num_lags > 0:
                Z = Z * self.gamma[None, None, :, None].astype(Z.dtype)
            Z = Z.reshape(lt, T, -1)
        return Z

    def _apply_scaling_to_incremental_tensors(self, Z):
        lt, T, nf = Z.shape[0], Z.shape[1], Z.shape[-1]
        if self.lengthscales is not None:                                                  # :391
            Z = Z.reshape(lt, T, 2, self.num_lags + 1, self.num_features)
            Z = Z / self.lengthscales[None, None, None, None, :].astype(Z.dtype)
            if self.num_lags > 0:
                Z = Z * self.gamma[None, None, None, :, None].astype(Z.dtype)
        return Z.reshape(lt, T, 2, nf)                                                     # :397

    # -- level tensors ---------------------------------------------------------------------
    def _levels(self, M):
        if self.order == 1:                                                                # :200,:232
            return signature_kern_first_order(M, self.num_levels, difference=self.difference)
        return signature_kern_higher_order(M, self.num_levels, order=self.order, difference=self.difference)

    def _K_seq_diag(self, X):
        """kernels.py:188-205.  X (N, L, d) -> (M+1, N)."""
        return self._levels(self._base_kern(X))                                            # :198 (batched)

    def _K_seq(self, X, X2=None):
        """kernels.py:208-237.  -> (M+1, N1, N2)."""
        N, L, d = X.shape
        if X2 is None:
            Xf = X.reshape(N * L, d)
            M = self._base_kern(Xf).reshape(N, L, N, L)                                    # :225-226
        else:
            N2, L2 = X2.shape[0], X2.shape[1]
            M = self._base_kern(X.reshape(N * L, d), X2.reshape(N2 * L2, d)).reshape(N, L, N2, L2)  # :228-230
        return self._levels(M)

    def _K_tens(self, Z, increments=False):
        """kernels.py:263-283.  -> (M+1, T, T)."""
        lt, T, nf = Z.shape[0], Z.shape[1], Z.shape[-1]
        if increments:
            Zf = Z.reshape(lt, 2 * T, nf)                                                  # :275
            M = self._base_kern(Zf).reshape(lt, T, 2, T, 2)                                # :276
            M = M[:, :, 1, :, 1] + M[:, :, 0, :, 0] - M[:, :, 1, :, 0] - M[:, :, 0, :, 1]  # :277
        else:
            M = self._base_kern(Z)                                                         # :279
        return tensor_kern(M, self.num_levels)

    def _K_tens_vs_seq(self, Z, X, increments=False):
        """kernels.py:313-340.  -> (M+1, T, N)."""
        lt, T, nf = Z.shape[0], Z.shape[1], Z.shape[-1]
        N, L = X.shape[0], X.shape[1]
        Xf = X.reshape(N * L, nf)                                                          # :326
        if increments:
            M = self._base_kern(Z.reshape(2 * T * lt, nf), Xf).reshape(lt, T, 2, N, L)     # :328-329
            M = M[:, :, 1] - M[:, :, 0]                                                    # :330
        else:
            M = self._base_kern(Z.reshape(T * lt, nf), Xf).reshape(lt, T, N, L)            # :332-333
        if self.order == 1:
            return signature_kern_tens_vs_seq_first_order(M, self.num_levels, difference=self.difference)
        return signature_kern_tens_vs_seq_higher_order(M, self.num_levels, order=self.order, difference=self.difference)

    def _weights(self):
        return (self.sigma * self.variances).astype(self.dtype)

    def _seq3(self, X):
        X = np.asarray(X, dtype=self.dtype)
        return X.reshape(X.shape[0], -1, self.num_features)                                # :417-418

    # -- public surface --------------------------------------------------------------------
    def K(self, X, X2=None, return_levels=False):
        """kernels.py:401-476 (exact branch)."""
        X = self._seq3(X)
        N = X.shape[0]
        Xs = self._apply_scaling_and_lags_to_sequences(X)                                  # :421
        if X2 is None:
            K = self._K_seq(Xs)                                                            # :428
            if self.normalization:
                K = K + JITTER * np.eye(N, dtype=self.dtype)[None]                         # :431
                dsq = np.sqrt(np.diagonal(K, axis1=1, axis2=2))                            # :432
                K = K / (dsq[:, :, None] * dsq[:, None, :])                                # :433
        else:
            X2 = self._seq3(X2)
            X2s = self._apply_scaling_and_lags_to_sequences(X2)                            # :440
            K = self._K_seq(Xs, X2s)                                                       # :453
            if self.normalization:
                d1 = np.sqrt(self._K_seq_diag(Xs) + JITTER)                                # :460-466
                d2 = np.sqrt(self._K_seq_diag(X2s) + JITTER)
                K = K / (d1[:, :, None] * d2[:, None, :])                                  # :469
        K = K * self._weights()[:, None, None]                                             # :471
        return K if return_levels else K.sum(axis=0)                                       # :473-476

    def Kdiag(self, X, return_levels=False):
        """kernels.py:479-510."""
        X = np.asarray(X, dtype=self.dtype)
        N = X.shape[0]
        if self.normalization:                                                             # :486-490
            if return_levels:
                return np.tile(self._weights()[:, None], [1, N])
            return np.full((N,), self.sigma * np.sum(self.variances), dtype=self.dtype)
        Xs = self._apply_scaling_and_lags_to_sequences(self._seq3(X))                      # :495-497
        Kd = self._K_seq_diag(Xs) * self._weights()[:, None]                               # :503-505
        return Kd if return_levels else Kd.sum(axis=0)

    def _scale_Z(self, Z, increments):
        Z = np.asarray(Z, dtype=self.dtype)
        return self._apply_scaling_to_incremental_tensors(Z) if increments else self._apply_scaling_to_tensors(Z)

    def K_tens(self, Z, return_levels=False, increments=False):
        """kernels.py:513-536.  Never normalised."""
        K = self._K_tens(self._scale_Z(Z, increments), increments) * self._weights()[:, None, None]  # :529-531
        return K if return_levels else K.sum(axis=0)

    def K_tens_vs_seq(self, Z, X, return_levels=False, increments=False):
        """kernels.py:539-588.  Normalised on the sequence axis only (:572-581)."""
        Xs = self._apply_scaling_and_lags_to_sequences(self._seq3(X))                      # :558
        K = self._K_tens_vs_seq(self._scale_Z(Z, increments), Xs, increments)              # :570
        if self.normalization:
            K = K / np.sqrt(self._K_seq_diag(Xs) + JITTER)[:, None, :]                     # :576-581
        K = K * self._weights()[:, None, None]                                             # :583
        return K if return_levels else K.sum(axis=0)

    def K_tens_n_seq_covs(self, Z, X, full_X_cov=False, return_levels=False, increments=False):
        """kernels.py:591-671."""
        Xs = self._apply_scaling_and_lags_to_sequences(self._seq3(X))                      # :610
        N = Xs.shape[0]
        Zs = self._scale_Z(Z, increments)
        Kzz = self._K_tens(Zs, increments)                                                 # :623
        Kzx = self._K_tens_vs_seq(Zs, Xs, increments)                                      # :624
        w = self._weights()
        if full_X_cov:
            Kxx = self._K_seq(Xs)                                                          # :630
            if self.normalization:
                Kxx = Kxx + JITTER * np.eye(N, dtype=self.dtype)[None]                     # :633
                dsq = np.sqrt(np.diagonal(Kxx, axis1=1, axis2=2))                          # :635
                Kxx = Kxx / (dsq[:, :, None] * dsq[:, None, :])                            # :637
                Kzx = Kzx / dsq[:, None, :]                                                # :638
            Kxx = Kxx * w[:, None, None]                                                   # :640
        else:
            Kxx = self._K_seq_diag(Xs)                                                     # :653
            if self.normalization:
                Kzx = Kzx / np.sqrt(Kxx + JITTER)[:, None, :]                              # :656-660
                Kxx = np.tile(w[:, None], [1, N])                                          # :661
            else:
                Kxx = Kxx * w[:, None]                                                     # :663
        Kzz = Kzz * w[:, None, None]                                                       # :641/:665
        Kzx = Kzx * w[:, None, None]                                                       # :642/:666
        if return_levels:
            return Kzz, Kzx, Kxx
        return Kzz.sum(axis=0), Kzx.sum(axis=0), Kxx.sum(axis=0)

    def K_seq_n_seq_covs(self, X, X2, full_X2_cov=False, return_levels=False):
        """kernels.py:674-761.  ``X`` = inducing sequences (already 3-D in the reference), ``X2`` = data."""
        Xs = self._apply_scaling_and_lags_to_sequences(self._seq3(X))                      # :690
        X2s = self._apply_scaling_and_lags_to_sequences(self._seq3(X2))                    # :691
        N, N2 = Xs.shape[0], X2s.shape[0]
        w = self._weights()
        Kxx = self._K_seq(Xs)                                                              # :704
        Kxx2 = self._K_seq(Xs, X2s)                                                        # :705
        if self.normalization:
            Kxx = Kxx + JITTER * np.eye(N, dtype=self.dtype)[None]                         # :709
            dsq = np.sqrt(np.diagonal(Kxx, axis1=1, axis2=2))                              # :711
            Kxx = Kxx / (dsq[:, :, None] * dsq[:, None, :])                                # :712
            Kxx2 = Kxx2 / dsq[:, :, None]                                                  # :713
        if full_X2_cov:
            Kx2x2 = self._K_seq(X2s)                                                       # :719
            if self.normalization:
                # Q6: the reference references undefined names here (:723-728); the evident intent
                # (mirror of :709-713) is implemented instead.
                Kx2x2 = Kx2x2 + JITTER * np.eye(N2, dtype=self.dtype)[None]
                d2 = np.sqrt(np.diagonal(Kx2x2, axis1=1, axis2=2))
                Kxx2 = Kxx2 / d2[:, None, :]
                Kx2x2 = Kx2x2 / (d2[:, :, None] * d2[:, None, :])
            Kx2x2 = Kx2x2 * w[:, None, None]                                               # :732
        else:
            Kx2x2 = self._K_seq_diag(X2s)                                                  # :743
            if self.normalization:
                d2 = np.sqrt(Kx2x2 + JITTER)                                               # :746-748
                # NOTE (reference quirk, reproduced): :750 divides Kxx2 by dsq[:, :, None] a SECOND time
                # (it was already divided at :713).
                Kxx2 = Kxx2 / (dsq[:, :, None] * d2[:, None, :])                           # :750
                Kx2x2 = np.tile(w[:, None], [1, N2])                                       # :751
            else:
                Kx2x2 = Kx2x2 * w[:, None]                                                 # :753
        Kxx = Kxx * w[:, None, None]                                                       # :730/:755
        Kxx2 = Kxx2 * w[:, None, None]                                                     # :731/:756
        if return_levels:
            return Kxx, Kxx2, Kx2x2
        return Kxx.sum(axis=0), Kxx2.sum(axis=0), Kx2x2.sum(axis=0)

    # numpy-facing wrappers (kernels.py:141-186)
    def compute_K(self, X, Y): return self.K(X, Y)
    def compute_K_symm(self, X): return self.K(X)
    def compute_K_level_diags(self, X): return self.Kdiag(X, return_levels=True)
    def compute_K_levels(self, X, X2): return self.K(X, X2, return_levels=True)
    def compute_Kdiag(self, X): return self.Kdiag(X)
    def compute_K_tens(self, Z): return self.K_tens(Z)
    def compute_K_tens_vs_seq(self, Z, X): return self.K_tens_vs_seq(Z, X)
    def compute_K_incr_tens(self, Z): return self.K_tens(Z, increments=True)
    def compute_K_incr_tens_vs_seq(self, Z, X): return self.K_tens_vs_seq(Z, X, increments=True)

    def compute_base_kern_symm(self, X):
        """kernels.py:150-157."""
        X = self._seq3(X)
        N, L, _ = X.shape
        Xs = self._apply_scaling_and_lags_to_sequences(X).reshape(N * L, -1)
        return self._base_kern(Xs).reshape(N, L, N, L).transpose(0, 2, 1, 3)


# ---------------------------------------------------------------------------
# gpsig/inducing_variables.py
# ---------------------------------------------------------------------------
def _mix_levels(W, Kl, two_sided):
    """Kzz[0] + sum_m W_m Kzz_m W_m^T   (inducing_variables.py:56, :83, :106)  or
    Kzx[0] + sum_m W_m Kzx_m         (:57, :73, :117)."""
    if two_sided:
        return Kl[0] + np.sum(np.matmul(np.matmul(W, Kl[1:]), np.swapaxes(W, -1, -2)), axis=0)
    return Kl[0] + np.sum(np.matmul(W, Kl[1:]), axis=0)


def inducing_tensors_Kuu(kern, Z, increments=False, W=None, jitter=0.0):
    """inducing_variables.py:78-87."""
    if W is not None:
        Kzz = _mix_levels(W, kern.K_tens(Z, return_levels=True, increments=increments), True)
    else:
        Kzz = kern.K_tens(Z, increments=increments)
    return Kzz + jitter * np.eye(Z.shape[1], dtype=Kzz.dtype)


def inducing_tensors_Kuf(kern, Z, X, increments=False, W=None):
    """inducing_variables.py:68-76."""
    if W is not None:
        return _mix_levels(W, kern.K_tens_vs_seq(Z, X, return_levels=True, increments=increments), False)
    return kern.K_tens_vs_seq(Z, X, increments=increments)


def inducing_tensors_Kuu_Kuf_Kff(kern, Z, X, increments=False, W=None, jitter=0.0, full_f_cov=False):
    """inducing_variables.py:51-66.  Q6: ``tf.shape(X)`` at :63 is an undefined name; X_new is meant."""
    if W is not None:
        Kzz, Kzx, Kxx = kern.K_tens_n_seq_covs(Z, X, full_X_cov=full_f_cov, return_levels=True, increments=increments)
        Kzz, Kzx, Kxx = _mix_levels(W, Kzz, True), _mix_levels(W, Kzx, False), Kxx.sum(axis=0)
    else:
        Kzz, Kzx, Kxx = kern.K_tens_n_seq_covs(Z, X, full_X_cov=full_f_cov, increments=increments)
    Kzz = Kzz + jitter * np.eye(Z.shape[1], dtype=Kzz.dtype)
    Kxx = Kxx + (jitter * np.eye(np.asarray(X).shape[0], dtype=Kxx.dtype) if full_f_cov else jitter)
    return Kzz, Kzx, Kxx


def inducing_sequences_Kuu(kern, Z, W=None, jitter=0.0):
    """inducing_variables.py:101-110."""
    if W is not None:
        Kzz = _mix_levels(W, kern.K(Z, return_levels=True), True)
    else:
        Kzz = kern.K(Z)
    return Kzz + jitter * np.eye(Z.shape[0], dtype=Kzz.dtype)


def inducing_sequences_Kuf(kern, Z, X, W=None):
    """inducing_variables.py:112-120."""
    if W is not None:
        return _mix_levels(W, kern.K(Z, X, return_levels=True), False)
    return kern.K(Z, X)


def inducing_sequences_Kuu_Kuf_Kff(kern, Z, X, W=None, jitter=0.0, full_f_cov=False):
    """inducing_variables.py:122-137."""
    if W is not None:
        Kzz, Kzx, Kxx = kern.K_seq_n_seq_covs(Z, X, full_X2_cov=full_f_cov, return_levels=True)
        Kzz, Kzx, Kxx = _mix_levels(W, Kzz, True), _mix_levels(W, Kzx, False), Kxx.sum(axis=0)
    else:
        Kzz, Kzx, Kxx = kern.K_seq_n_seq_covs(Z, X, full_X2_cov=full_f_cov)
    Kzz = Kzz + jitter * np.eye(np.asarray(Z).shape[0], dtype=Kzz.dtype)
    Kxx = Kxx + (jitter * np.eye(np.asarray(X).shape[0], dtype=Kxx.dtype) if full_f_cov else jitter)
    return Kzz, Kzx, Kxx


# ---------------------------------------------------------------------------
# Low-rank algorithms: gpsig/low_rank_calculations.py, gpsig/signature_algs.py:162-222, gpsig/kernels.py:239-311.
# The reference draws landmarks and projections with TensorFlow's RNG inside the graph (not reproducible), so these
# restatements take the random objects as ARGUMENTS: `landmarks` (c, d') and `sketches` (one per level >= 2: plain data
# -- r outputs, column pointers colptr (r+1), coordinate pairs i1 / i2 and values val (nnz) -- applied by apply_sketch
# below; no code of the product is called).  Parity with the reference is therefore statistical only;
# parity between the HIP path and this restatement, given the same random objects, is exact.
# Q6 (reference bugs NOT reproduced): signature_algs.py:191 appends reduce_sum(U) instead of reduce_sum(P);
# kernels.py:425,448-449 pass the unscaled X to the low-rank feature map while Kdiag (:500) passes the scaled one.
# ---------------------------------------------------------------------------
def apply_sketch(sk, A, B):
    """lr_hadamard_prod_rand (low_rank_calculations.py:76-193) given its random matrix: output column j of the projected
    row-wise Kronecker product is  sum_e val[e] * A[..., i1[e]] * B[..., i2[e]]  over the entries e of column j
    ('lin', :104-127: one entry per column with a Rademacher sign; 'sqrt' / 'log', :152-193: the non-zeros of the very
    sparse Gaussian matrix, already scaled by sqrt(s / r)).  (..., k1), (..., k2) -> (..., r)."""
    colptr, i1, i2, val = (np.asarray(getattr(sk, n)) for n in ("colptr", "i1", "i2", "val"))
    out = np.zeros(A.shape[:-1] + (int(sk.r),), dtype=np.result_type(A, B))
    for j in range(int(sk.r)):
        e = slice(int(colptr[j]), int(colptr[j + 1]))
        out[..., j] = np.einsum("...e,...e,e->...", A[..., i1[e]], B[..., i2[e]], val[e])
    return out


def nystrom_whitening(kern_fn, landmarks, jitter_diag):
    """low_rank_calculations.py:50-57: W = k(S,S) + diag(jitter_diag); eig; S += jitter; returns U / sqrt(S) (c, c).
    An eigensolver fixes each eigenvector up to sign only (tf.self_adjoint_eig at :55 as much as LAPACK here), and the
    level >= 2 features depend on those signs through the random projections of coordinate pairs; to make an evaluation a
    function of its random objects the component of largest magnitude of every eigenvector is taken positive -- a
    convention, not something the reference states."""
    W = kern_fn(landmarks, landmarks) + np.diag(jitter_diag)                      # :51-52
    S, U = np.linalg.eigh(W)                                                     # :55
    top = np.argmax(np.abs(U), axis=0)
    U = U * np.where(U[top, np.arange(U.shape[1])] < 0, -1.0, 1.0)[None, :]
    S = S + JITTER                                                               # :56
    return U / np.sqrt(S)[None, :]                                               # :57, :60


def nystrom_map(X, kern_fn, landmarks, whitening):
    """low_rank_calculations.py:59-61: k(X, S) @ U / D.  X: (P, d') -> (P, c)."""
    return kern_fn(X, landmarks) @ whitening


def signature_kern_first_order_lr_feature(U, num_levels, sketches, difference=True):
    """signature_algs.py:162-192 with the evident fix of :191.  U: (N, L, c) -> list of (N, .) factors."""
    Phi = [np.ones((U.shape[0], 1), dtype=U.dtype)]                              # :177
    if difference:
        U = U[:, 1:, :] - U[:, :-1, :]                                           # :180
    Phi.append(U.sum(axis=1))                                                    # :182
    P = U                                                                        # :184
    for i in range(2, num_levels + 1):                                           # :185
        P = _excumsum(P, 1)                                                      # :186
        P = apply_sketch(sketches[i - 2], U, P)                                         # :188/:190  lr_hadamard_prod_rand(U, P, ...)
        Phi.append(P.sum(axis=1))                                                # :191 (Q6: the reference sums U here)
    return Phi


def tensor_kern_lr_feature(U, num_levels, sketches):
    """signature_algs.py:194-222.  U: (lt, T, c) -> list of (T, .) factors."""
    Phi = [np.ones((U.shape[1], 1), dtype=U.dtype)]                              # :209
    k = 0
    for i in range(1, num_levels + 1):                                           # :212
        R = U[k]; k += 1                                                         # :213-214
        for j in range(1, i):                                                    # :215
            R = apply_sketch(sketches[j - 1], U[k], R); k += 1                           # :217/:219  lr_hadamard_prod_rand(U[k], R, ...)
        Phi.append(R)                                                            # :221
    return Phi


class LowRankOracle:
    """The low_rank=True branches of SignatureKernel (kernels.py:424-426, 442-458, 499-501, 525-527, 560-574, 612-628)
    on top of a SignatureKernelOracle `k` (which supplies scaling, base kernel, weights)."""

    def __init__(self, k, landmarks, jitter_diag, sketches):
        self.k, self.S, self.sk = k, np.asarray(landmarks, dtype=np.float64), sketches
        self.Wh = nystrom_whitening(k._base_kern, self.S, jitter_diag)

    def seq_features(self, X):
        k = self.k
        Xs = k._apply_scaling_and_lags_to_sequences(k._seq3(X))                  # (Q6: scaled, as Kdiag :497-500 does)
        N, L, dd = Xs.shape
        F = nystrom_map(Xs.reshape(N * L, dd), k._base_kern, self.S, self.Wh).reshape(N, L, -1)    # kernels.py:252-254
        return signature_kern_first_order_lr_feature(F, k.num_levels, self.sk, difference=k.difference)   # :257

    def tens_features(self, Z, increments=False):
        k = self.k
        Zs = k._scale_Z(Z, increments)
        lt, T, dd = Zs.shape[0], Zs.shape[1], Zs.shape[-1]
        if increments:                                                           # kernels.py:300-304
            F = nystrom_map(Zs.reshape(lt * T * 2, dd), k._base_kern, self.S, self.Wh).reshape(lt, T, 2, -1)
            F = F[:, :, 1, :] - F[:, :, 0, :]
        else:                                                                    # :306-308
            F = nystrom_map(Zs.reshape(lt * T, dd), k._base_kern, self.S, self.Wh).reshape(lt, T, -1)
        return tensor_kern_lr_feature(F, k.num_levels, self.sk)                  # :310

    def _finish(self, Kl, return_levels):
        Kl = Kl * self.k._weights()[:, None, None]
        return Kl if return_levels else Kl.sum(axis=0)

    def K(self, X, X2=None, return_levels=False):
        """kernels.py:423-476, low-rank branch."""
        P1 = self.seq_features(X)
        if X2 is None:
            Kl = np.stack([P @ P.T for P in P1], axis=0)                          # :426
            if self.k.normalization:
                Kl = Kl + JITTER * np.eye(Kl.shape[1])[None]                      # :431
                dsq = np.sqrt(np.diagonal(Kl, axis1=1, axis2=2))
                Kl = Kl / (dsq[:, :, None] * dsq[:, None, :])                     # :432-433
        else:
            P2 = self.seq_features(X2)
            Kl = np.stack([a @ b.T for a, b in zip(P1, P2)], axis=0)              # :451
            if self.k.normalization:
                d1 = np.sqrt(np.stack([np.sum(np.square(P), axis=-1) for P in P1], axis=0) + JITTER)   # :457, :463
                d2 = np.sqrt(np.stack([np.sum(np.square(P), axis=-1) for P in P2], axis=0) + JITTER)
                Kl = Kl / (d1[:, :, None] * d2[:, None, :])                       # :469
        return self._finish(Kl, return_levels)

    def Kdiag(self, X, return_levels=False):
        """kernels.py:479-510."""
        k = self.k
        N = np.asarray(X).shape[0]
        if k.normalization:
            return np.tile(k._weights()[:, None], [1, N]) if return_levels else np.full((N,), k.sigma * np.sum(k.variances))
        Kd = np.stack([np.sum(np.square(P), axis=-1) for P in self.seq_features(X)], axis=0) * k._weights()[:, None]   # :501-505
        return Kd if return_levels else Kd.sum(axis=0)

    def K_tens(self, Z, return_levels=False, increments=False):
        """kernels.py:525-536."""
        Kl = np.stack([P @ P.T for P in self.tens_features(Z, increments)], axis=0)
        return self._finish(Kl, return_levels)

    def K_tens_vs_seq(self, Z, X, return_levels=False, increments=False):
        """kernels.py:560-588."""
        PZ, PX = self.tens_features(Z, increments), self.seq_features(X)
        Kl = np.stack([a @ b.T for a, b in zip(PZ, PX)], axis=0)                  # :568
        if self.k.normalization:
            dx = np.sqrt(np.stack([np.sum(np.square(P), axis=-1) for P in PX], axis=0) + JITTER)       # :574-580
            Kl = Kl / dx[:, None, :]                                              # :581
        return self._finish(Kl, return_levels)


    def K_seq_n_seq_covs(self, X, X2, full_X2_cov=False, return_levels=False):
        """kernels.py:674-761, low-rank branch (:696-703, :718, :740); normalisation as in the exact branch, including the
        double division of :713 + :750 and the intent of the undefined names at :723-728."""
        k = self.k
        P1, P2 = self.seq_features(X), self.seq_features(X2)
        N, N2 = P1[0].shape[0], P2[0].shape[0]
        w = k._weights()
        Kxx = np.stack([P @ P.T for P in P1], axis=0)                             # :702
        Kxx2 = np.stack([a @ b.T for a, b in zip(P1, P2)], axis=0)                # :703
        if k.normalization:
            Kxx = Kxx + JITTER * np.eye(N)[None]                                  # :709
            dsq = np.sqrt(np.diagonal(Kxx, axis1=1, axis2=2))
            Kxx = Kxx / (dsq[:, :, None] * dsq[:, None, :])
            Kxx2 = Kxx2 / dsq[:, :, None]                                         # :713
        if full_X2_cov:
            K22 = np.stack([P @ P.T for P in P2], axis=0)                         # :718
            if k.normalization:
                K22 = K22 + JITTER * np.eye(N2)[None]
                d2 = np.sqrt(np.diagonal(K22, axis1=1, axis2=2))
                Kxx2 = Kxx2 / d2[:, None, :]
                K22 = K22 / (d2[:, :, None] * d2[:, None, :])
            K22 = K22 * w[:, None, None]
        else:
            K22 = np.stack([np.sum(np.square(P), axis=-1) for P in P2], axis=0)   # :740
            if k.normalization:
                d2 = np.sqrt(K22 + JITTER)
                Kxx2 = Kxx2 / (dsq[:, :, None] * d2[:, None, :])                  # :750
                K22 = np.tile(w[:, None], [1, N2])
            else:
                K22 = K22 * w[:, None]
        Kxx, Kxx2 = Kxx * w[:, None, None], Kxx2 * w[:, None, None]
        if return_levels:
            return Kxx, Kxx2, K22
        return Kxx.sum(axis=0), Kxx2.sum(axis=0), K22.sum(axis=0)


# ---------------------------------------------------------------------------
# Independent validators (NOT restatements of the reference): they stand in for esig, which the
# reference's notebook uses and this image lacks.
# ---------------------------------------------------------------------------
def truncated_signature(x, num_levels):
    """Signature of the piecewise-linear path through the rows of ``x`` (L, d), truncated at
    ``num_levels``; flattened to (1 + d + ... + d^M,) exactly as ``esig.tosig.stream2sig`` lays it out
    (notebook cell 6-7).  Chen's identity: S = prod_t exp(dx_t), exp(v) = sum_m v^{(x)m}/m!."""
    x = np.asarray(x, dtype=np.float64)
    d = x.shape[1]
    sig = [np.ones(())] + [np.zeros((d,) * m) for m in range(1, num_levels + 1)]
    for t in range(x.shape[0] - 1):
        v = x[t + 1] - x[t]
        ex = [np.ones(())]
        for m in range(1, num_levels + 1):
            ex.append(np.multiply.outer(ex[-1], v) / m)
        new = []
        for m in range(num_levels + 1):
            acc = np.zeros((d,) * m)
            for k in range(m + 1):
                acc = acc + np.multiply.outer(sig[k], ex[m - k])
            new.append(acc)
        sig = new
    return np.concatenate([s.reshape(-1) for s in sig])


def signature_level_slices(d, num_levels):
    out, start = [], 0
    for m in range(num_levels + 1):
        out.append(slice(start, start + d ** m))
        start += d ** m
    return out


def rank1_tensor_features(Z, num_levels):
    """Explicit flattened rank-1 tensors from components Z (lt, T, d): the construction of notebook
    cell 18 (level m = z_{m,1} (x) ... (x) z_{m,m}, first component varying slowest)."""
    Z = np.asarray(Z, dtype=np.float64)
    T = Z.shape[1]
    feats, k = [np.ones((T, 1))], 0
    for m in range(1, num_levels + 1):
        Zm = Z[k]; k += 1
        for _ in range(1, m):
            Zm = (Zm[..., None] * Z[k, :, None, :]).reshape(T, -1); k += 1
        feats.append(Zm)
    return np.concatenate(feats, axis=1)


def brute_force_first_order(dM, num_levels):
    """K_m = sum over strictly increasing index tuples a_1<...<a_m, b_1<...<b_m of prod dM[a_l, b_l]
    for ONE pair's increment lattice dM (l1, l2).  Pure-Python loops: small cases only."""
    from itertools import combinations
    l1, l2 = dM.shape
    out = [1.0]
    for m in range(1, num_levels + 1):
        tot = 0.0
        for A in combinations(range(l1), m):
            for B in combinations(range(l2), m):
                p = 1.0
                for a, b in zip(A, B):
                    p *= dM[a, b]
                tot += p
        out.append(tot)
    return np.asarray(out)


# ---------------------------------------------------------------------------
# CPU baseline helper for bench.py (op-for-op graph, tiled so that it fits in memory)
# ---------------------------------------------------------------------------
def K_symm_tiled(kern: SignatureKernelOracle, X, tile=64):
    """`kern.K(X)` evaluated on tiles of `tile` x `tile` sequences so the (N*L)^2 base-kernel tensor
    of kernels.py:226 never has to exist at once (it is 550 GB at N=4096, L=64).  Same ops per tile."""
    X = kern._seq3(X)
    N = X.shape[0]
    Xs = kern._apply_scaling_and_lags_to_sequences(X)
    Kl = np.empty((kern.num_levels + 1, N, N), dtype=kern.dtype)
    for i0 in range(0, N, tile):
        for j0 in range(0, N, tile):
            Kl[:, i0:i0 + tile, j0:j0 + tile] = kern._K_seq(Xs[i0:i0 + tile], Xs[j0:j0 + tile])
    if kern.normalization:
        Kl = Kl + JITTER * np.eye(N, dtype=kern.dtype)[None]
        dsq = np.sqrt(np.diagonal(Kl, axis1=1, axis2=2))
        Kl = Kl / (dsq[:, :, None] * dsq[:, None, :])
    Kl = Kl * kern._weights()[:, None, None]
    return Kl.sum(axis=0)
