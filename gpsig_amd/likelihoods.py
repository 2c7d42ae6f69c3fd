"""Likelihoods for the SVGP built on the signature covariances (torch).

The reference uses GPflow 1.5.1's likelihood objects (requirements.txt:8, not vendored): ``Bernoulli`` and ``MultiClass``
in benchmarks/models/train_gpsig.py:61-64, ``Gaussian`` in GPflow's regression examples.  Their published algorithms are
restated here: ``variational_expectations`` (the data-fit term of the ELBO, gpsig/models.py:54) and ``predict_mean_and_var``.
Gauss-Hermite quadrature with 20 points, as GPflow's ``num_gauss_hermite_points`` default.
"""
import math

import numpy as np
import torch

NUM_GH = 20


_GH_CACHE = {}


def _gh(dtype, device, n=NUM_GH):
    """Gauss-Hermite nodes and weights on the device, uploaded once per (dtype, device) -- a host-to-device copy per call would
    also keep a training step from being recorded as a HIP graph."""
    key = (dtype, str(device), n)
    if key not in _GH_CACHE:
        x, w = np.polynomial.hermite.hermgauss(n)
        _GH_CACHE[key] = (torch.as_tensor(x, dtype=dtype, device=device), torch.as_tensor(w, dtype=dtype, device=device))
    return _GH_CACHE[key]


def inv_probit(x):
    """gpflow.likelihoods.inv_probit: Phi(x) squashed into [1e-3, 1 - 1e-3]."""
    jitter = 1e-3
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) * (1 - 2 * jitter) + jitter


class Gaussian(torch.nn.Module):
    """gpflow.likelihoods.Gaussian: y = f + N(0, variance)."""

    def __init__(self, variance=1.0, device="cuda"):
        super().__init__()
        from .autodiff import positive_inverse
        self.raw_variance = torch.nn.Parameter(torch.as_tensor(positive_inverse(variance), dtype=torch.float64, device=device))

    @property
    def variance(self):
        from .autodiff import positive
        return positive(self.raw_variance)

    def variational_expectations(self, Fmu, Fvar, Y):
        v = self.variance
        return -0.5 * math.log(2 * math.pi) - 0.5 * torch.log(v) - 0.5 * ((Y - Fmu) ** 2 + Fvar) / v

    def predict_mean_and_var(self, Fmu, Fvar):
        return Fmu, Fvar + self.variance


class Bernoulli(torch.nn.Module):
    """gpflow.likelihoods.Bernoulli with the probit link (the default), Y in {0, 1}."""

    def variational_expectations(self, Fmu, Fvar, Y):
        x, w = _gh(Fmu.dtype, Fmu.device)
        F = Fmu[..., None] + torch.sqrt(2.0 * Fvar)[..., None] * x            # ndiagquad: (N, R, H)
        p = inv_probit(F)
        logp = torch.log(torch.where(Y[..., None] == 1, p, 1 - p))            # logdensities.bernoulli
        return (logp * (w / math.sqrt(math.pi))).sum(-1)

    def predict_mean_and_var(self, Fmu, Fvar):
        p = inv_probit(Fmu / torch.sqrt(1 + Fvar))
        return p, p - p * p


class MultiClass(torch.nn.Module):
    """gpflow.likelihoods.MultiClass with the RobustMax inverse link (epsilon = 1e-3): Y (N, 1) holds class indices and
    there is one latent function per class."""

    def __init__(self, num_classes, epsilon=1e-3):
        super().__init__()
        self.num_classes, self.epsilon = int(num_classes), float(epsilon)
        self.eps_k1 = self.epsilon / (self.num_classes - 1.0)

    def prob_is_largest(self, Y, mu, var):
        """RobustMax.prob_is_largest: probability that the latent of the observed class is the largest."""
        x, w = _gh(mu.dtype, mu.device)
        oh_on = torch.nn.functional.one_hot(Y.reshape(-1).long(), self.num_classes).to(mu.dtype)        # (N, K)
        mu_sel = (oh_on * mu).sum(1)
        var_sel = (oh_on * var).sum(1)
        X = mu_sel[:, None] + x[None, :] * torch.sqrt(torch.clamp(2.0 * var_sel, min=1e-10))[:, None]   # (N, H)
        dist = (X[:, None, :] - mu[:, :, None]) / torch.sqrt(torch.clamp(var, min=1e-10))[:, :, None]   # (N, K, H)
        cdfs = 0.5 * (1.0 + torch.erf(dist / math.sqrt(2.0)))
        cdfs = cdfs * (1 - 2e-4) + 1e-4
        oh_off = 1.0 - oh_on
        cdfs = cdfs * oh_off[:, :, None] + oh_on[:, :, None]
        # product over the classes as exp(sum(log)): every factor lies in [1e-4, 1], and torch.prod's backward inspects its
        # input for zeros on the host (a stream synchronisation per step, and nothing a HIP-graph capture can contain)
        prod = torch.exp(torch.log(cdfs).sum(dim=1))
        return (prod @ (w / math.sqrt(math.pi))[:, None])                                               # (N, 1)

    def variational_expectations(self, Fmu, Fvar, Y):
        p = self.prob_is_largest(Y, Fmu, Fvar)
        return p * math.log(1.0 - self.epsilon) + (1.0 - p) * math.log(self.eps_k1)

    def predict_mean_and_var(self, Fmu, Fvar):
        """Class probabilities p(y = k) for every k (GPflow's predict_mean_and_var of MultiClass)."""
        N = Fmu.shape[0]
        ps = []
        for k in range(self.num_classes):
            Yk = torch.full((N, 1), k, dtype=torch.long, device=Fmu.device)
            p = self.prob_is_largest(Yk, Fmu, Fvar)
            ps.append(p * (1 - self.epsilon) + (1.0 - p) * self.eps_k1)
        ps = torch.cat(ps, dim=1)
        return ps, ps - ps * ps
