"""Differentiable CPU oracle: the reference's graph restated in torch (float64), differentiated by torch.autograd.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as ``sigkern_oracle.py``: only ``tests/`` and
``__graft_entry__.smoke()`` may import it).

The reference has no gradient code of its own -- it is trained by TensorFlow's reverse-mode autodiff of the graph
built by ``gpsig/signature_algs.py`` / ``gpsig/kernels.py`` (``training.py:149-164``).  The gradients the HIP path has
to reproduce are therefore "autodiff of that graph".  This file restates the graph, operation for operation, in
torch (each function cites the reference lines it follows, paths relative to ``/root/reference``) and lets
``torch.autograd`` play TensorFlow's role.  It is pinned in two ways (``tests/test_oracle_torch.py``): its VALUES must
equal the NumPy oracle's on the same inputs, and its GRADIENTS must pass ``torch.autograd.gradcheck``-style finite
differences of the NumPy oracle.  First- and higher-order algorithms, exact (non low-rank) branches.
"""
from __future__ import annotations

import math

import torch

JITTER = 1e-6  # gpflow.settings.jitter


def _excumsum(A, dim):
    """tf.cumsum(A, exclusive=True, axis=dim)."""
    return torch.cumsum(A, dim=dim) - A


def signature_kern_first_order(M, num_levels, difference=True):
    """gpsig/signature_algs.py:8-35.  M (N1, L1, N2, L2) or (N, L, L)."""
    if M.dim() == 4:
        K = [torch.ones((M.shape[0], M.shape[2]), dtype=M.dtype)]                                   # :19-20
    else:
        K = [torch.ones((M.shape[0],), dtype=M.dtype)]                                              # :22-23
    if difference:
        M = M[:, 1:, ..., 1:] + M[:, :-1, ..., :-1] - M[:, :-1, ..., 1:] - M[:, 1:, ..., :-1]      # :26
    K.append(M.sum(dim=(1, -1)))                                                                    # :28
    R = M
    for _ in range(2, num_levels + 1):                                                              # :31
        R = M * _excumsum(_excumsum(R, 1), -1)                                                      # :32
        K.append(R.sum(dim=(1, -1)))                                                                # :33
    return torch.stack(K, dim=0)


def signature_kern_higher_order(M, num_levels, order=2, difference=True):
    """gpsig/signature_algs.py:37-74.  M (N1, L1, N2, L2) or (N, L, L)."""
    if M.dim() == 4:
        K = [torch.ones((M.shape[0], M.shape[2]), dtype=M.dtype)]                                   # :48-50
    else:
        K = [torch.ones((M.shape[0],), dtype=M.dtype)]                                              # :52-53
    if difference:
        M = M[:, 1:, ..., 1:] + M[:, :-1, ..., :-1] - M[:, :-1, ..., 1:] - M[:, 1:, ..., :-1]      # :56
    K.append(M.sum(dim=(1, -1)))                                                                    # :58
    R = [[M]]                                                                                       # :60
    for i in range(2, num_levels + 1):                                                              # :61
        d = min(i, order)                                                                           # :62
        dp = len(R)
        Rn = [[None] * d for _ in range(d)]
        Rn[0][0] = M * _excumsum(_excumsum(sum(R[r][s] for r in range(dp) for s in range(dp)), 1), -1)            # :64
        for j in range(2, d + 1):                                                                   # :65
            Rn[0][j - 1] = 1.0 / j * M * _excumsum(sum(R[r][j - 2] for r in range(dp)), 1)          # :66
            Rn[j - 1][0] = 1.0 / j * M * _excumsum(sum(R[j - 2][s] for s in range(dp)), -1)         # :67
            for k in range(2, d + 1):                                                               # :68
                Rn[j - 1][k - 1] = 1.0 / (j * k) * M * R[j - 2][k - 2]                              # :69
        K.append(sum(Rn[r][s] for r in range(d) for s in range(d)).sum(dim=(1, -1)))                # :71
        R = Rn                                                                                      # :72
    return torch.stack(K, dim=0)


def signature_kern_tens_vs_seq_higher_order(M, num_levels, order=2, difference=True):
    """gpsig/signature_algs.py:129-160.  M (lt, T, N, L)."""
    if difference:
        M = M[..., 1:] - M[..., :-1]                                                                # :142
    K = [torch.ones(M.shape[1:3], dtype=M.dtype)]                                                   # :144
    k = 0
    for i in range(1, num_levels + 1):                                                              # :147
        R = [M[k]]; k += 1                                                                          # :148-149
        for j in range(1, i):                                                                       # :150
            d = min(j + 1, order)                                                                   # :151
            Rn = [M[k] * _excumsum(sum(R), 2)]                                                      # :153
            for l in range(1, d):                                                                   # :154
                Rn.append(1.0 / (l + 1) * M[k] * R[l - 1])                                          # :155
            R = Rn; k += 1                                                                          # :156-157
        K.append(sum(R).sum(dim=2))                                                                 # :158
    return torch.stack(K, dim=0)


def tensor_kern(M, num_levels):
    """gpsig/signature_algs.py:76-99.  M (lt, T, T')."""
    K = [torch.ones(M.shape[1:], dtype=M.dtype)]
    k = 0
    for i in range(1, num_levels + 1):
        R = M[k]; k += 1
        for _ in range(1, i):
            R = M[k] * R; k += 1
        K.append(R)
    return torch.stack(K, dim=0)


def signature_kern_tens_vs_seq_first_order(M, num_levels, difference=True):
    """gpsig/signature_algs.py:101-127.  M (lt, T, N, L)."""
    if difference:
        M = M[..., 1:] - M[..., :-1]                                                                # :114
    K = [torch.ones(M.shape[1:3], dtype=M.dtype)]
    k = 0
    for i in range(1, num_levels + 1):
        R = M[k]; k += 1
        for _ in range(1, i):
            R = M[k] * _excumsum(R, 2); k += 1                                                      # :123-124
        K.append(R.sum(dim=2))                                                                      # :125
    return torch.stack(K, dim=0)


# ---- gpsig/lags.py ----------------------------------------------------------------------------------------------------
def lin_interp(time, X, time_query):
    """gpsig/lags.py:7-38 (3-D branch).  X (N, L, d), time (L,), time_query (L, p) -> (N, L, p, d)."""
    dist = time[:, None, None] - time_query[None, :, :]                                             # :20
    masked = torch.where(dist > JITTER, torch.full_like(dist, -math.inf), dist)                     # :22
    left = torch.argmax(masked, dim=0)
    right = torch.clamp(left + 1, max=X.shape[1] - 1)
    Xl, Xr = X[:, left, :], X[:, right, :]
    tl, tr = time[left], time[right]
    return Xl + (time_query[None, ..., None] - tl[None, ..., None]) * (Xr - Xl) / (tr[None, ..., None] - tl[None, ..., None])


def add_lags_to_sequences(X, lags):
    """gpsig/lags.py:41-63."""
    L = X.shape[1]
    time = torch.arange(L, dtype=X.dtype) / (L - 1)
    time_lags = torch.clamp(time[:, None] - lags[None, :], min=0.)                                  # :57
    X_lags = lin_interp(time, X, time_lags)
    return torch.cat((X[:, :, None, :], X_lags), dim=2)


# ---- base kernels (gpsig/kernels.py:765-781, 799-993) -----------------------------------------------------------------
def _mm_t(X, X2):
    return torch.matmul(X, X2.transpose(-1, -2))


def _square_dist(X, X2=None):
    Xs = torch.sum(torch.square(X), dim=-1)
    if X2 is None:
        return -2 * _mm_t(X, X) + Xs[..., :, None] + Xs[..., None, :]
    X2s = torch.sum(torch.square(X2), dim=-1)
    return -2 * _mm_t(X, X2) + Xs[..., :, None] + X2s[..., None, :]


def _euclid_dist(X, X2=None):
    return torch.sqrt(torch.clamp(_square_dist(X, X2), min=1e-40))                                 # tf.maximum(r2, 1e-40)


def base_spectral(X, X2, alpha, omega, gamma, family):
    """gpsig/kernels.py:921-942 (2-D or batched inputs; 'mixed' as evidently intended: the reference's branch has undefined names and a
    sign slip, :932-936 -- the first floor(Q/2) components Gaussian, the rest exponential).  The square root of :924 is taken with
    derivative 0 at 0: TensorFlow's is inf there, which makes every gradient of K(X, X) NaN in the reference itself."""
    X2 = X if X2 is None else X2
    diff = X[..., :, None, :] - X2[..., None, :, :]                                                 # :923 / :925 (one copy per q below)
    Q = alpha.shape[0]
    out = 0.0
    for q in range(Q):
        sq = torch.sum(torch.square(diff * gamma[q]), dim=-1)
        if family == "rbf" or (family == "mixed" and q < Q // 2):
            env = torch.exp(-sq / 2)                                                                # :928
        else:
            root = torch.where(sq > 0, torch.sqrt(torch.where(sq > 0, sq, torch.ones_like(sq))), torch.zeros_like(sq))
            env = torch.exp(-root / 2)                                                              # :926
        out = out + alpha[q] * env * torch.cos(2. * math.pi * torch.sum(diff * omega[q], dim=-1))   # :937, :942
    return out


def base_kernel(name, X, X2=None, p0=None, p1=None, spectral=None):
    if name == "spectral":
        return base_spectral(X, X2, *spectral)
    if name == "linear":
        return _mm_t(X, X if X2 is None else X2)
    if name == "cosine":
        Xn = torch.sqrt(torch.sum(torch.square(X), dim=-1))
        X2n = Xn if X2 is None else torch.sqrt(torch.sum(torch.square(X2), dim=-1))
        return _mm_t(X, X if X2 is None else X2) / (Xn[..., :, None] * X2n[..., None, :])
    if name == "poly":
        return (_mm_t(X, X if X2 is None else X2) + p0) ** p1
    if name == "rbf":
        return torch.exp(-_square_dist(X, X2) / 2)
    if name == "mix":
        inner = _mm_t(X, X if X2 is None else X2)
        return p0 * torch.exp(-_square_dist(X, X2) / 2) + (1. - p0) * inner
    r = _euclid_dist(X, X2)
    if name == "matern12":
        return torch.exp(-r)
    if name == "matern32":
        return (1. + math.sqrt(3.) * r) * torch.exp(-math.sqrt(3.) * r)
    if name == "matern52":
        return (1.0 + math.sqrt(5.) * r + 5. / 3. * torch.square(r)) * torch.exp(-math.sqrt(5.) * r)
    raise ValueError(name)


class SignatureKernelTorchOracle:
    """torch restatement of ``gpsig.kernels.SignatureKernel`` (exact branches, order 1); every hyper-parameter is a
    float64 tensor that may require grad: variances (M+1,), sigma (), lengthscales (d,) or None, lags (p,), gamma (p+1,),
    p0 (base-kernel parameter: gamma of poly, mixing of mix)."""

    def __init__(self, num_features, num_levels, base="linear", variances=None, sigma=1.0, lengthscales=None, normalization=True,
                 difference=True, num_lags=0, lags=None, gamma=None, p0=None, p1=None, order=1, spectral=None):
        t = lambda v: v if isinstance(v, torch.Tensor) else torch.as_tensor(v, dtype=torch.float64)
        self.num_features, self.num_levels, self.base = num_features, num_levels, base
        self.normalization, self.difference, self.num_lags = normalization, difference, num_lags
        self.variances = t(torch.ones(num_levels + 1, dtype=torch.float64) if variances is None else variances)
        self.sigma = t(sigma)
        self.lengthscales = None if lengthscales is None else t(lengthscales)
        self.lags = None if lags is None else t(lags)
        self.gamma = None if gamma is None else t(gamma)
        self.p0 = None if p0 is None else t(p0)
        self.p1 = p1
        self.spectral = spectral                                                                    # (alpha (Q,), omega (Q, d), gamma (Q, d), family)
        self.order = num_levels if (order <= 0 or order >= num_levels) else order                   # kernels.py:57

    def _base(self, X, X2=None):
        return base_kernel(self.base, X, X2, self.p0, self.p1, self.spectral)

    def scale_sequences(self, X):
        """kernels.py:343-364.  X (N, L, d) -> (N, L, d*(num_lags+1))."""
        N, L, _ = X.shape
        if self.num_lags > 0:
            X = add_lags_to_sequences(X, self.lags)
        X = X.reshape(N, L, self.num_lags + 1, self.num_features)
        if self.lengthscales is not None:
            X = X / self.lengthscales[None, None, None, :]
        if self.num_lags > 0:
            X = X * self.gamma[None, None, :, None]
        return X.reshape(N, L, -1)

    def scale_tensors(self, Z, increments):
        """kernels.py:367-398."""
        lt, T = Z.shape[0], Z.shape[1]
        if self.lengthscales is not None:
            shape = Z.shape
            Z = Z.reshape(*shape[:-1], self.num_lags + 1, self.num_features) / self.lengthscales
            if self.num_lags > 0:
                Z = Z * self.gamma[:, None]
            Z = Z.reshape(shape)
        return Z

    # level primitives on scaled inputs
    def _seq_alg(self, M):
        if self.order == 1:                                                                         # kernels.py:201-204, :233-236
            return signature_kern_first_order(M, self.num_levels, self.difference)
        return signature_kern_higher_order(M, self.num_levels, self.order, self.difference)

    def K_seq_diag_levels(self, Xs):
        return self._seq_alg(self._base(Xs))                                                        # kernels.py:188-205

    def K_seq_levels(self, Xs, X2s=None):
        N, L, d = Xs.shape                                                                          # kernels.py:208-237
        if X2s is None:
            M = self._base(Xs.reshape(N * L, d)).reshape(N, L, N, L)
        else:
            N2, L2 = X2s.shape[:2]
            M = self._base(Xs.reshape(N * L, d), X2s.reshape(N2 * L2, d)).reshape(N, L, N2, L2)
        return self._seq_alg(M)

    def K_tens_levels(self, Zs, increments):
        lt, T, nf = Zs.shape[0], Zs.shape[1], Zs.shape[-1]                                          # kernels.py:263-283
        if increments:
            M = self._base(Zs.reshape(lt, 2 * T, nf)).reshape(lt, T, 2, T, 2)
            M = M[:, :, 1, :, 1] + M[:, :, 0, :, 0] - M[:, :, 1, :, 0] - M[:, :, 0, :, 1]
        else:
            M = self._base(Zs)
        return tensor_kern(M, self.num_levels)

    def K_tens_vs_seq_levels(self, Zs, Xs, increments):
        lt, T, nf = Zs.shape[0], Zs.shape[1], Zs.shape[-1]                                          # kernels.py:313-340
        N, L = Xs.shape[:2]
        Xf = Xs.reshape(N * L, nf)
        if increments:
            M = self._base(Zs.reshape(2 * T * lt, nf), Xf).reshape(lt, T, 2, N, L)
            M = M[:, :, 1] - M[:, :, 0]
        else:
            M = self._base(Zs.reshape(T * lt, nf), Xf).reshape(lt, T, N, L)
        if self.order == 1:                                                                         # kernels.py:336-339
            return signature_kern_tens_vs_seq_first_order(M, self.num_levels, self.difference)
        return signature_kern_tens_vs_seq_higher_order(M, self.num_levels, self.order, self.difference)

    def _w(self):
        return self.sigma * self.variances

    def _seq3(self, X):
        return X.reshape(X.shape[0], -1, self.num_features)

    # public surface
    def K(self, X, X2=None, return_levels=False):
        """kernels.py:401-476."""
        Xs = self.scale_sequences(self._seq3(X))
        N = Xs.shape[0]
        if X2 is None:
            K = self.K_seq_levels(Xs)
            if self.normalization:
                K = K + JITTER * torch.eye(N, dtype=K.dtype)[None]
                dsq = torch.sqrt(torch.diagonal(K, dim1=1, dim2=2))
                K = K / (dsq[:, :, None] * dsq[:, None, :])
        else:
            X2s = self.scale_sequences(self._seq3(X2))
            K = self.K_seq_levels(Xs, X2s)
            if self.normalization:
                d1 = torch.sqrt(self.K_seq_diag_levels(Xs) + JITTER)
                d2 = torch.sqrt(self.K_seq_diag_levels(X2s) + JITTER)
                K = K / (d1[:, :, None] * d2[:, None, :])
        K = K * self._w()[:, None, None]
        return K if return_levels else K.sum(dim=0)

    def Kdiag(self, X, return_levels=False):
        """kernels.py:479-510."""
        N = X.shape[0]
        if self.normalization:
            Kd = self._w()[:, None].repeat(1, N)
        else:
            Kd = self.K_seq_diag_levels(self.scale_sequences(self._seq3(X))) * self._w()[:, None]
        return Kd if return_levels else Kd.sum(dim=0)

    def K_tens(self, Z, return_levels=False, increments=False):
        """kernels.py:513-536."""
        K = self.K_tens_levels(self.scale_tensors(Z, increments), increments) * self._w()[:, None, None]
        return K if return_levels else K.sum(dim=0)

    def K_tens_vs_seq(self, Z, X, return_levels=False, increments=False):
        """kernels.py:539-588."""
        Xs = self.scale_sequences(self._seq3(X))
        K = self.K_tens_vs_seq_levels(self.scale_tensors(Z, increments), Xs, increments)
        if self.normalization:
            K = K / torch.sqrt(self.K_seq_diag_levels(Xs) + JITTER)[:, None, :]
        K = K * self._w()[:, None, None]
        return K if return_levels else K.sum(dim=0)

    def K_tens_n_seq_covs(self, Z, X, full_X_cov=False, increments=False):
        """kernels.py:591-671 (summed levels)."""
        Xs = self.scale_sequences(self._seq3(X))
        N = Xs.shape[0]
        Zs = self.scale_tensors(Z, increments)
        Kzz = self.K_tens_levels(Zs, increments)
        Kzx = self.K_tens_vs_seq_levels(Zs, Xs, increments)
        w = self._w()
        if full_X_cov:
            Kxx = self.K_seq_levels(Xs)
            if self.normalization:
                Kxx = Kxx + JITTER * torch.eye(N, dtype=Kxx.dtype)[None]
                dsq = torch.sqrt(torch.diagonal(Kxx, dim1=1, dim2=2))
                Kxx = Kxx / (dsq[:, :, None] * dsq[:, None, :])
                Kzx = Kzx / dsq[:, None, :]
            Kxx = Kxx * w[:, None, None]
        else:
            Kxx = self.K_seq_diag_levels(Xs)
            if self.normalization:
                Kzx = Kzx / torch.sqrt(Kxx + JITTER)[:, None, :]
                Kxx = w[:, None].repeat(1, N)
            else:
                Kxx = Kxx * w[:, None]
        Kzz = Kzz * w[:, None, None]
        Kzx = Kzx * w[:, None, None]
        return Kzz.sum(dim=0), Kzx.sum(dim=0), Kxx.sum(dim=0)


# ---------------------------------------------------------------------------
# Low-rank mode (gpsig/low_rank_calculations.py, gpsig/signature_algs.py:162-222, gpsig/kernels.py:239-311 and the low_rank
# branches of K / K_tens / K_tens_vs_seq / K_tens_n_seq_covs), differentiable: what TensorFlow's autodiff sees when the
# reference trains with low_rank=True (an option of benchmarks/models/train_gpsig.py:21, :58).  As in oracle/sigkern_oracle.py the random objects are
# ARGUMENTS -- here the landmark INDICES into the concatenation of scaled points the reference gathers from (tf.gather at
# kernels.py:446, :563, :615, :700: gradients flow into the landmarks), the jitter draw of low_rank_calculations.py:52 and one
# sparse projection per level >= 2 (plain data: r, colptr, i1, i2, val) -- and Q6 applies (scaled inputs; :191 sums P).
# Test infrastructure: pinned to the NumPy restatement by value in tests/test_grad_core.py.
# ---------------------------------------------------------------------------
def apply_sketch(sk, A, B):
    """out[..., j] = sum_{e in column j} val[e] A[..., i1[e]] B[..., i2[e]]   (low_rank_calculations.py:76-193 given the matrix)."""
    colptr = [int(v) for v in sk.colptr]
    i1 = torch.as_tensor([int(v) for v in sk.i1], dtype=torch.long)
    i2 = torch.as_tensor([int(v) for v in sk.i2], dtype=torch.long)
    val = torch.as_tensor([float(v) for v in sk.val], dtype=torch.float64)
    cols = []
    for j in range(int(sk.r)):
        e = slice(colptr[j], colptr[j + 1])
        cols.append((A[..., i1[e]] * B[..., i2[e]] * val[e]).sum(dim=-1))
    return torch.stack(cols, dim=-1)


class LowRankTorchOracle(SignatureKernelTorchOracle):
    """The low_rank=True branches on top of the exact restatement: the level primitives become products of low-rank factors, the
    normalisation / weighting code of the base class is what the reference shares between the two modes."""

    def set_draw(self, idx, jitter_diag, sketches):
        self._idx = torch.as_tensor([int(v) for v in idx], dtype=torch.long)
        self._jd = torch.as_tensor([float(v) for v in jitter_diag], dtype=torch.float64)
        self._sk = list(sketches)
        return self

    def _open(self, *points):
        pool = torch.cat([p.reshape(-1, p.shape[-1]) for p in points], dim=0)
        self._S = pool[self._idx]                                                                   # low_rank_calculations.py:47-48
        W = self._base(self._S, self._S) + torch.diag(self._jd)                                     # :51-52
        ev, U = torch.linalg.eigh(W)                                                                # :55
        top = U.detach().abs().argmax(dim=0)                                                        # sign convention of oracle/sigkern_oracle.py:nystrom_whitening
        sgn = torch.where(U.detach()[top, torch.arange(U.shape[1])] < 0, -1.0, 1.0).to(U.dtype)
        self._Wh = U * sgn[None, :] / torch.sqrt(ev + JITTER)[None, :]                              # :56-57, :60
        self._feat = {}

    def _nys(self, pts):
        return self._base(pts, self._S) @ self._Wh                                                  # :59-61

    def _seq_feat(self, Xs):
        if id(Xs) not in self._feat:
            N, L, d = Xs.shape
            U = self._nys(Xs.reshape(N * L, d)).reshape(N, L, -1)                                   # kernels.py:252-254
            if self.difference:
                U = U[:, 1:] - U[:, :-1]                                                            # signature_algs.py:180
            Phi = [torch.ones((N, 1), dtype=U.dtype), U.sum(dim=1)]                                 # :177, :182
            P = U
            for i in range(2, self.num_levels + 1):
                P = _excumsum(P, 1)                                                                 # :186
                P = apply_sketch(self._sk[i - 2], U, P)                                             # :188 / :190
                Phi.append(P.sum(dim=1))                                                            # :191 (Q6)
            self._feat[id(Xs)] = (Xs, Phi)
        return self._feat[id(Xs)][1]

    def _tens_feat(self, Zs, increments):
        if id(Zs) not in self._feat:
            lt, T, d = Zs.shape[0], Zs.shape[1], Zs.shape[-1]
            if increments:                                                                          # kernels.py:300-304
                F = self._nys(Zs.reshape(lt * T * 2, d)).reshape(lt, T, 2, -1)
                F = F[:, :, 1] - F[:, :, 0]
            else:
                F = self._nys(Zs.reshape(lt * T, d)).reshape(lt, T, -1)                             # :306-308
            Phi, k = [torch.ones((T, 1), dtype=F.dtype)], 0                                         # signature_algs.py:209
            for i in range(1, self.num_levels + 1):
                R = F[k]; k += 1
                for j in range(1, i):
                    R = apply_sketch(self._sk[j - 1], F[k], R); k += 1                              # :217 / :219
                Phi.append(R)
            self._feat[id(Zs)] = (Zs, Phi)
        return self._feat[id(Zs)][1]

    # level primitives as products of factors (kernels.py:426, :451, :457, :501, :527, :568)
    def K_seq_levels(self, Xs, X2s=None):
        P1 = self._seq_feat(Xs)
        P2 = P1 if X2s is None else self._seq_feat(X2s)
        return torch.stack([a @ b.T for a, b in zip(P1, P2)], dim=0)

    def K_seq_diag_levels(self, Xs):
        return torch.stack([torch.square(P).sum(dim=-1) for P in self._seq_feat(Xs)], dim=0)

    def K_tens_levels(self, Zs, increments):
        return torch.stack([P @ P.T for P in self._tens_feat(Zs, increments)], dim=0)

    def K_tens_vs_seq_levels(self, Zs, Xs, increments):
        return torch.stack([a @ b.T for a, b in zip(self._tens_feat(Zs, increments), self._seq_feat(Xs))], dim=0)

    # the public surface: gather the landmarks from this evaluation's points, then the shared code.  The base class scales its
    # inputs itself; scaling is deterministic, so the features are keyed by the tensors it hands to the primitives.
    def scale_sequences(self, X):
        key = ("s", id(X))
        if key not in self._scaled:
            self._scaled[key] = (X, super().scale_sequences(X))
        return self._scaled[key][1]

    def scale_tensors(self, Z, increments):
        key = ("t", id(Z))
        if key not in self._scaled:
            self._scaled[key] = (Z, super().scale_tensors(Z, increments))
        return self._scaled[key][1]

    def _seq3(self, X):
        key = ("3", id(X))
        if key not in self._scaled:
            self._scaled[key] = (X, super()._seq3(X))
        return self._scaled[key][1]

    def K(self, X, X2=None, return_levels=False):
        self._scaled = {}
        pts = [self.scale_sequences(self._seq3(X))] + ([] if X2 is None else [self.scale_sequences(self._seq3(X2))])
        self._open(*pts)                                                                            # kernels.py:445-446
        return super().K(X, X2, return_levels)

    def Kdiag(self, X, return_levels=False):
        self._scaled = {}
        if not self.normalization:
            self._open(self.scale_sequences(self._seq3(X)))
        return super().Kdiag(X, return_levels)

    def K_tens(self, Z, return_levels=False, increments=False):
        self._scaled = {}
        self._open(self.scale_tensors(Z, increments))
        return super().K_tens(Z, return_levels, increments)

    def K_tens_vs_seq(self, Z, X, return_levels=False, increments=False):
        self._scaled = {}
        self._open(self.scale_tensors(Z, increments), self.scale_sequences(self._seq3(X)))          # kernels.py:562-563
        return super().K_tens_vs_seq(Z, X, return_levels, increments)

    def K_tens_n_seq_covs(self, Z, X, full_X_cov=False, increments=False):
        self._scaled = {}
        self._open(self.scale_tensors(Z, increments), self.scale_sequences(self._seq3(X)))          # kernels.py:614-615
        return super().K_tens_n_seq_covs(Z, X, full_X_cov, increments)
