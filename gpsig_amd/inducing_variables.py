"""Inducing variables for signature kernels: the call surface of ``gpsig.inducing_variables``
(reference: gpsig/inducing_variables.py).  The reference registers these with GPflow's multiple
dispatch; here ``Kuu`` / ``Kuf`` / ``Kuu_Kuf_Kff`` are plain functions that switch on the feature class.
"""
import numpy as np

from .kernels import SignatureKernel, _is_torch

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class SignatureInducing:
    """Reference: inducing_variables.py:14-26.  ``W`` (num_levels, n, n) mixes the inducing variables
    level by level when ``learn_weights`` is set."""

    def __init__(self, Z, num_levels, learn_weights=False):
        self.Z = Z
        self.learn_weights = learn_weights
        if learn_weights:
            self.W = np.tile(np.eye(self.__len__())[None, ...], [num_levels, 1, 1])

    def __len__(self):
        return self.Z.shape[0]


class InducingTensors(SignatureInducing):
    """Reference: inducing_variables.py:28-49.  Z is (M(M+1)/2, num_tensors, d') or, with increments,
    (M(M+1)/2, num_tensors, 2, d')."""

    def __init__(self, Z, num_levels, increments=False, **kwargs):
        len_tensors = int(num_levels * (num_levels + 1) / 2)
        assert Z.shape[0] == len_tensors
        if increments:
            assert Z.ndim == 4
            assert Z.shape[2] == 2
        super().__init__(Z, num_levels, **kwargs)
        self.len_tensors = len_tensors
        self.increments = increments

    def __len__(self):
        return self.Z.shape[1]


class InducingSequences(SignatureInducing):
    """Reference: inducing_variables.py:89-98.  Z is (num_inducing, len_inducing, num_features)."""

    def __init__(self, Z, num_levels, **kwargs):
        super().__init__(Z, num_levels, **kwargs)
        self.len_inducing = Z.shape[1]


def _is_kernel(kern):
    # a SignatureKernel, or its trainable view gpsig_amd.autodiff.SignatureKernelModule (same methods, torch tensors)
    return isinstance(kern, SignatureKernel) or isinstance(getattr(kern, "kern", None), SignatureKernel)


def _mm(A, B):
    return torch.matmul(A, B) if _is_torch(A) or _is_torch(B) else np.matmul(A, B)


def _like(W, ref):
    if _is_torch(ref):
        return torch.as_tensor(W, dtype=ref.dtype, device=ref.device)
    return np.asarray(W)


def _t(A):
    return A.transpose(-1, -2) if _is_torch(A) else np.swapaxes(A, -1, -2)


def _mix_square(W, K):   # Kzz[0] + sum_m W_m Kzz_m W_m^T    (inducing_variables.py:56, :83, :106, :127)
    W = _like(W, K)
    return K[0] + _mm(_mm(W, K[1:]), _t(W)).sum(0)


def _mix_left(W, K):     # Kzx[0] + sum_m W_m Kzx_m          (inducing_variables.py:57, :73, :117, :128)
    W = _like(W, K)
    return K[0] + _mm(W, K[1:]).sum(0)


def _eye_like(n, ref):
    return torch.eye(n, dtype=ref.dtype, device=ref.device) if _is_torch(ref) else np.eye(n, dtype=ref.dtype)


def Kuu(feat, kern, *, jitter=0.0):
    """Reference: inducing_variables.py:78-87 (tensors), :101-110 (sequences)."""
    assert _is_kernel(kern)
    if isinstance(feat, InducingTensors):
        if feat.learn_weights:
            Kzz = _mix_square(feat.W, kern.K_tens(feat.Z, return_levels=True, increments=feat.increments))
        else:
            Kzz = kern.K_tens(feat.Z, increments=feat.increments)
    elif isinstance(feat, InducingSequences):
        if feat.learn_weights:
            Kzz = _mix_square(feat.W, kern.K(feat.Z, return_levels=True, presliced=True))
        else:
            Kzz = kern.K(feat.Z, presliced=True)
    else:
        raise NotImplementedError("Kuu for %s" % type(feat).__name__)
    return Kzz + jitter * _eye_like(len(feat), Kzz)


def Kuf(feat, kern, X_new):
    """Reference: inducing_variables.py:68-76 (tensors), :112-120 (sequences)."""
    assert _is_kernel(kern)
    if isinstance(feat, InducingTensors):
        if feat.learn_weights:
            return _mix_left(feat.W, kern.K_tens_vs_seq(feat.Z, X_new, return_levels=True, increments=feat.increments))
        return kern.K_tens_vs_seq(feat.Z, X_new, increments=feat.increments)
    if isinstance(feat, InducingSequences):
        if feat.learn_weights:
            return _mix_left(feat.W, kern.K(feat.Z, X_new, presliced_X=True, return_levels=True))
        return kern.K(feat.Z, X_new, presliced_X=True)
    raise NotImplementedError("Kuf for %s" % type(feat).__name__)


def Kuu_Kuf_Kff(feat, kern, X_new, *, jitter=0.0, full_f_cov=False):
    """Reference: inducing_variables.py:51-66 (tensors), :122-137 (sequences): the three matrices SVGP needs in
    one call.  (``tf.shape(X)`` at :63/:134 is an undefined name in the reference; X_new is what is meant.)"""
    assert _is_kernel(kern)
    if isinstance(feat, InducingTensors):
        if feat.learn_weights:
            Kzz, Kzx, Kxx = kern.K_tens_n_seq_covs(feat.Z, X_new, full_X_cov=full_f_cov, return_levels=True,
                                                   increments=feat.increments)
            Kzz, Kzx, Kxx = _mix_square(feat.W, Kzz), _mix_left(feat.W, Kzx), Kxx.sum(0)
        else:
            Kzz, Kzx, Kxx = kern.K_tens_n_seq_covs(feat.Z, X_new, full_X_cov=full_f_cov, increments=feat.increments)
    elif isinstance(feat, InducingSequences):
        if feat.learn_weights:
            Kzz, Kzx, Kxx = kern.K_seq_n_seq_covs(feat.Z, X_new, full_X2_cov=full_f_cov, return_levels=True)
            Kzz, Kzx, Kxx = _mix_square(feat.W, Kzz), _mix_left(feat.W, Kzx), Kxx.sum(0)
        else:
            Kzz, Kzx, Kxx = kern.K_seq_n_seq_covs(feat.Z, X_new, full_X2_cov=full_f_cov)
    else:
        raise NotImplementedError("Kuu_Kuf_Kff for %s" % type(feat).__name__)
    Kzz = Kzz + jitter * _eye_like(len(feat), Kzz)
    if full_f_cov:
        Kxx = Kxx + jitter * _eye_like(X_new.shape[0], Kxx)
    else:
        Kxx = Kxx + jitter
    return Kzz, Kzx, Kxx
