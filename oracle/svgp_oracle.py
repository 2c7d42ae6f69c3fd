"""CPU restatement (NumPy / SciPy) of the dense SVGP algebra the reference takes from GPflow 1.5.1
(requirements.txt:8; NOT under /root/reference, so restated from its published algorithm):
``gpflow.conditionals.base_conditional`` and ``gpflow.kullback_leiblers.gauss_kl`` as called by
gpsig/models.py:46-51, :62-73.  TEST INFRASTRUCTURE ONLY (see oracle/sigkern_oracle.py)."""
import numpy as np
from scipy.linalg import cholesky, solve_triangular


def base_conditional(Kmn, Kmm, Knn, f, full_cov=False, q_sqrt=None, white=False):
    R = f.shape[1]
    Lm = cholesky(Kmm, lower=True)
    A = solve_triangular(Lm, Kmn, lower=True)
    fvar = np.tile((Knn - A.T @ A)[None], [R, 1, 1]) if full_cov else np.tile((Knn - np.sum(A * A, 0))[None], [R, 1])
    if not white:
        A = solve_triangular(Lm.T, A, lower=False)
    fmean = A.T @ f
    if q_sqrt is not None:
        if q_sqrt.ndim == 2:
            LTA = A[None] * q_sqrt.T[:, :, None]
        else:
            LTA = np.matmul(np.swapaxes(np.tril(q_sqrt), 1, 2), A[None])
        fvar = fvar + (np.matmul(np.swapaxes(LTA, 1, 2), LTA) if full_cov else np.sum(LTA * LTA, 1))
    return fmean, (fvar if full_cov else fvar.T)


def gauss_kl(q_mu, q_sqrt, K=None):
    M, R = q_mu.shape
    white, diag = K is None, q_sqrt.ndim == 2
    if white:
        alpha = q_mu
    else:
        Lp = cholesky(K, lower=True)
        alpha = solve_triangular(Lp, q_mu, lower=True)
    Lq = q_sqrt if diag else np.tril(q_sqrt)
    Lq_diag = q_sqrt if diag else np.diagonal(Lq, axis1=1, axis2=2)
    two_kl = np.sum(alpha ** 2) - R * M - np.sum(np.log(Lq_diag ** 2))
    if white:
        two_kl += np.sum(Lq ** 2)
    else:
        if diag:
            Kinv = np.linalg.inv(K)
            two_kl += np.sum(np.diag(Kinv)[:, None] * q_sqrt ** 2)
        else:
            for r in range(R):
                two_kl += np.sum(solve_triangular(Lp, Lq[r], lower=True) ** 2)
        two_kl += R * np.sum(np.log(np.diag(Lp) ** 2))
    return 0.5 * two_kl


# ---- likelihoods (GPflow 1.5.1 gpflow/likelihoods.py, restated; 20-point Gauss-Hermite as GPflow's default) ----------
def _gh(n=20):
    return np.polynomial.hermite.hermgauss(n)


def gaussian_variational_expectations(Fmu, Fvar, Y, variance):
    return -0.5 * np.log(2 * np.pi) - 0.5 * np.log(variance) - 0.5 * ((Y - Fmu) ** 2 + Fvar) / variance


def _inv_probit(x):
    from scipy.special import erf
    return 0.5 * (1.0 + erf(x / np.sqrt(2.0))) * (1 - 2e-3) + 1e-3


def bernoulli_variational_expectations(Fmu, Fvar, Y):
    x, w = _gh()
    out = np.zeros_like(Fmu)
    for xi, wi in zip(x, w):
        p = _inv_probit(Fmu + np.sqrt(2.0 * Fvar) * xi)
        out += wi / np.sqrt(np.pi) * np.log(np.where(Y == 1, p, 1 - p))
    return out


def multiclass_variational_expectations(Fmu, Fvar, Y, num_classes, epsilon=1e-3):
    from scipy.special import erf
    x, w = _gh()
    N = Fmu.shape[0]
    p = np.zeros((N, 1))
    y = Y.reshape(-1).astype(int)
    for n in range(N):
        acc = 0.0
        for xi, wi in zip(x, w):
            X = Fmu[n, y[n]] + xi * np.sqrt(max(2.0 * Fvar[n, y[n]], 1e-10))
            prod = 1.0
            for k in range(num_classes):
                if k == y[n]:
                    continue
                cdf = 0.5 * (1.0 + erf((X - Fmu[n, k]) / np.sqrt(max(Fvar[n, k], 1e-10)) / np.sqrt(2.0)))
                prod *= cdf * (1 - 2e-4) + 1e-4
            acc += wi / np.sqrt(np.pi) * prod
        p[n, 0] = acc
    return p * np.log(1.0 - epsilon) + (1.0 - p) * np.log(epsilon / (num_classes - 1.0))
