"""gpsig_amd.likelihoods (plain torch, device-agnostic) against the NumPy restatement in oracle/svgp_oracle.py."""
import numpy as np
import torch

from gpsig_amd import likelihoods as LK
from oracle import svgp_oracle as SO


def test_variational_expectations_match_the_oracle():
    rng = np.random.default_rng(0)
    N, K = 17, 4
    Fmu, Fvar = rng.standard_normal((N, K)), rng.uniform(0.05, 2.0, (N, K))
    t = torch.tensor
    # Gaussian
    Y = rng.standard_normal((N, K))
    g = LK.Gaussian(variance=0.3, device="cpu")
    assert np.allclose(g.variational_expectations(t(Fmu), t(Fvar), t(Y)).detach().numpy(),
                       SO.gaussian_variational_expectations(Fmu, Fvar, Y, 0.3), rtol=1e-12, atol=1e-12)
    # Bernoulli
    Yb = rng.integers(0, 2, (N, 1)).astype(np.float64)
    got = LK.Bernoulli().variational_expectations(t(Fmu[:, :1]), t(Fvar[:, :1]), t(Yb)).numpy()
    assert np.allclose(got, SO.bernoulli_variational_expectations(Fmu[:, :1], Fvar[:, :1], Yb), rtol=1e-12, atol=1e-12)
    # MultiClass (RobustMax)
    Yc = rng.integers(0, K, (N, 1))
    got = LK.MultiClass(K).variational_expectations(t(Fmu), t(Fvar), t(Yc)).numpy()
    assert np.allclose(got, SO.multiclass_variational_expectations(Fmu, Fvar, Yc, K), rtol=1e-11, atol=1e-12)
    ps, _ = LK.MultiClass(K).predict_mean_and_var(t(Fmu), t(Fvar))
    assert ps.shape == (N, K) and float(ps.min()) > 0 and np.all(np.abs(ps.sum(1).numpy() - 1) < 0.05)
