"""gpsig_amd -- MI355X-native signature-kernel evaluation (the hot path of tgcsaba/GPSig).

    from gpsig_amd import kernels, inducing_variables
    kern = kernels.SignatureRBF(input_dim=L * d, num_features=d, num_levels=4)
    K = kern.compute_K_symm(X)            # X: (N, L*d) float64 numpy array (or a CUDA torch tensor)

Everything is computed by hand-written HIP kernels for gfx950 behind the C ABI of include/gpsig_hip.h;
there is no CPU fallback."""
from . import _lib, inducing_variables, kernels, utils  # noqa: F401

__all__ = ["kernels", "inducing_variables", "utils"]   # training side: gpsig_amd.autodiff, gpsig_amd.models, gpsig_amd.likelihoods
