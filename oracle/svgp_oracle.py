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
