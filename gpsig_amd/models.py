"""Sparse variational GP prediction on top of the signature covariances: the call surface of ``gpsig.models.SVGP``'s
``_build_predict`` (reference: gpsig/models.py:62-73) and the prior KL of ``_build_likelihood`` (:46-51).

The three covariances come from ``inducing_variables.Kuu_Kuf_Kff`` (HIP kernels); the dense linear algebra --
Cholesky of Kzz, the triangular solves, the matmuls of GPflow 1.5.1's ``conditionals.base_conditional`` and
``kullback_leiblers.gauss_kl`` (pinned by requirements.txt:8, not vendored; restated from their published
algorithm) -- runs on the device through torch.linalg, i.e. rocSOLVER (potrf, trsm) and rocBLAS / hipBLASLt.
``SVGPModule`` is the trainable model: the ELBO of ``_build_likelihood`` (models.py:40-59) on top of
``gpsig_amd.autodiff.SignatureKernelModule`` (HIP forward and backward kernels) and ``gpsig_amd.likelihoods``;
``fit`` is a plain Adam loop (the reference drives TF optimisers through GPflow actions, gpsig/training.py --
control plane, not rebuilt).
"""
import copy
import numpy as np

from . import inducing_variables as iv

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

JITTER = 1e-6   # gpflow.settings.jitter (models.py:65)


def _dev(a, device):
    if torch.is_tensor(a):
        return a.to(device=device, dtype=torch.float64)
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=device)


def _cholesky(K):
    """rocSOLVER potrf.  torch.linalg.cholesky reads the factorisation's status back to the host to raise on failure, which a
    stream capture cannot contain: while a HIP graph of the training step is being recorded the unchecked form is used (the
    eager warm-up iterations before a capture run the checked one on the same matrices)."""
    if K.is_cuda and torch.cuda.is_current_stream_capturing():
        return torch.linalg.cholesky_ex(K, check_errors=False).L
    return torch.linalg.cholesky(K)


def base_conditional(Kmn, Kmm, Knn, f, *, full_cov=False, q_sqrt=None, white=False):
    """GPflow 1.5.1 ``conditionals.base_conditional``.  Kmn (M, N), Kmm (M, M), Knn (N,) or (N, N), f (M, R),
    q_sqrt None, (M, R) [diagonal] or (R, M, M) [lower triangular].  Returns fmean (N, R), fvar (N, R) or (R, N, N)."""
    R = f.shape[1]
    Lm = _cholesky(Kmm)                                                       # rocSOLVER potrf
    A = torch.linalg.solve_triangular(Lm, Kmn, upper=False)                   # trsm: Lm^-1 Kmn
    if full_cov:
        fvar = (Knn - A.T @ A).unsqueeze(0).repeat(R, 1, 1)
    else:
        fvar = (Knn - (A * A).sum(0)).unsqueeze(0).repeat(R, 1)
    if not white:
        A = torch.linalg.solve_triangular(Lm.T, A, upper=True)                # Lm^-T A
    fmean = A.T @ f
    if q_sqrt is not None:
        if q_sqrt.dim() == 2:
            LTA = A.unsqueeze(0) * q_sqrt.T.unsqueeze(2)                      # (R, M, N)
        elif q_sqrt.dim() == 3:
            LTA = torch.tril(q_sqrt).transpose(1, 2) @ A.unsqueeze(0)
        else:
            raise ValueError("Bad dimension for q_sqrt: %s" % str(q_sqrt.dim()))
        fvar = fvar + (LTA.transpose(1, 2) @ LTA if full_cov else (LTA * LTA).sum(1))
    if not full_cov:
        fvar = fvar.T
    return fmean, fvar


def gauss_kl(q_mu, q_sqrt, K=None):
    """GPflow 1.5.1 ``kullback_leiblers.gauss_kl``: KL[N(q_mu, q_sqrt q_sqrt^T) || N(0, K)] summed over the R latent
    functions (K None: the prior is white).  q_mu (M, R); q_sqrt (M, R) diagonal or (R, M, M) lower triangular."""
    white = K is None
    diag = q_sqrt.dim() == 2
    M, R = q_mu.shape
    if white:
        alpha = q_mu
    else:
        Lp = _cholesky(K)
        alpha = torch.linalg.solve_triangular(Lp, q_mu, upper=False)
    if diag:
        Lq_diag = q_sqrt
    else:
        Lq = torch.tril(q_sqrt)
        Lq_diag = torch.diagonal(Lq, dim1=1, dim2=2)
    two_kl = (alpha * alpha).sum() - R * M - torch.log(Lq_diag * Lq_diag).sum()
    if white:
        two_kl = two_kl + ((q_sqrt * q_sqrt).sum() if diag else (Lq * Lq).sum())
    else:
        if diag:
            Lp_inv = torch.linalg.solve_triangular(Lp, torch.eye(M, dtype=K.dtype, device=K.device), upper=False)
            K_inv_diag = (Lp_inv * Lp_inv).sum(0)
            two_kl = two_kl + (K_inv_diag.unsqueeze(1) * q_sqrt * q_sqrt).sum()
        else:
            LpiLq = torch.linalg.solve_triangular(Lp.unsqueeze(0).expand(R, M, M), Lq, upper=False)
            two_kl = two_kl + (LpiLq * LpiLq).sum()
        two_kl = two_kl + R * torch.log(torch.diagonal(Lp) ** 2).sum()
    return 0.5 * two_kl


class SVGP:
    """Prediction half of ``gpsig.models.SVGP`` (gpsig/models.py:13-73): variational parameters + feature + kernel.

    :kern:     a gpsig_amd.kernels.SignatureKernel
    :feat:     InducingTensors or InducingSequences (models.py:19-20)
    :q_mu:     (num_inducing, num_latent); :q_sqrt: (num_latent, M, M) lower-triangular or (M, num_latent) if q_diag
    :whiten:   models.py:35 (default True)
    """

    def __init__(self, kern, feat, num_latent=1, q_diag=False, whiten=True, q_mu=None, q_sqrt=None, mean_function=None, device="cuda:0"):
        if not isinstance(feat, (iv.InducingTensors, iv.InducingSequences)):
            raise ValueError('feat must be of type either InducingTensors or InducingSequences')     # models.py:19-20
        self.kern, self.feature, self.q_diag, self.whiten = kern, feat, q_diag, whiten
        self.mean_function = mean_function
        self.device = torch.device(device)
        m = len(feat)
        self.num_latent = num_latent if q_mu is None else np.asarray(q_mu).shape[1]
        # GPflow's _init_variational_parameters: zero mean, identity square root
        self.q_mu = np.zeros((m, self.num_latent)) if q_mu is None else q_mu
        if q_sqrt is None:
            q_sqrt = np.ones((m, self.num_latent)) if q_diag else np.tile(np.eye(m)[None], [self.num_latent, 1, 1])
        self.q_sqrt = q_sqrt

    def _covs(self, X_new, full_cov):
        Kzz, Kzx, Kxx = iv.Kuu_Kuf_Kff(self.feature, self.kern, X_new, jitter=JITTER, full_f_cov=full_cov)   # models.py:65
        return _dev(Kzz, self.device), _dev(Kzx, self.device), _dev(Kxx, self.device)

    def predict_f(self, X_new, full_cov=False, return_Kzz=False):
        """models.py:62-73.  Returns (f_mean (N, R), f_var (N, R) or (R, N, N)) as torch tensors on the device."""
        Kzz, Kzx, Kxx = self._covs(X_new, full_cov)
        q_mu, q_sqrt = _dev(self.q_mu, self.device), _dev(self.q_sqrt, self.device)
        if not self.q_diag:
            q_sqrt = torch.tril(q_sqrt)                                                             # models.py:66
        f_mean, f_var = base_conditional(Kzx, Kzz, Kxx, q_mu, full_cov=full_cov, q_sqrt=q_sqrt, white=self.whiten)
        if self.mean_function is not None:
            f_mean = f_mean + _dev(self.mean_function(X_new), self.device)                           # models.py:67
        return (f_mean, f_var, Kzz) if return_Kzz else (f_mean, f_var)

    def prior_kl(self):
        """The KL term of models.py:46-51."""
        q_mu, q_sqrt = _dev(self.q_mu, self.device), _dev(self.q_sqrt, self.device)
        if not self.q_diag:
            q_sqrt = torch.tril(q_sqrt)
        if self.whiten:
            return gauss_kl(q_mu, q_sqrt)
        Kzz = _dev(iv.Kuu(self.feature, self.kern, jitter=JITTER), self.device)
        return gauss_kl(q_mu, q_sqrt, K=Kzz)


class SVGPModule(torch.nn.Module if torch is not None else object):
    """Trainable ``gpsig.models.SVGP`` (gpsig/models.py:13-73): kernel hyper-parameters, inducing tensors / sequences,
    optional level weights and the variational parameters are ``torch.nn.Parameter``s on the GPU.

    :kern:        a ``gpsig_amd.kernels.SignatureKernel`` (wrapped) or a ``gpsig_amd.autodiff.SignatureKernelModule``
    :feat:        ``InducingTensors`` / ``InducingSequences`` holding the initial Z (and ``learn_weights``)
    :likelihood:  an object of ``gpsig_amd.likelihoods``
    :num_data:    size of the full data set, for minibatch scaling of the data-fit term (models.py:57-58)
    """

    def __init__(self, kern, feat, likelihood, num_latent=1, q_diag=False, whiten=True, num_data=None, mean_function=None, device="cuda:0"):
        super().__init__()
        from .autodiff import SignatureKernelModule
        if not isinstance(feat, (iv.InducingTensors, iv.InducingSequences)):
            raise ValueError('feat must be of type either InducingTensors or InducingSequences')     # models.py:19-20
        dev = torch.device(device)
        self.kernel = kern if isinstance(kern, SignatureKernelModule) else SignatureKernelModule(kern, device=dev)
        self.likelihood = likelihood
        self.q_diag, self.whiten, self.num_data, self.mean_function = q_diag, whiten, num_data, mean_function
        self._feat_cls = type(feat)
        self._increments = bool(getattr(feat, "increments", False))
        self._learn_weights = bool(feat.learn_weights)
        self._num_levels = self.kernel.kern.num_levels
        self.Z = torch.nn.Parameter(_dev(feat.Z, dev))
        if self._learn_weights:
            self.W = torch.nn.Parameter(_dev(feat.W, dev))
        m = len(feat)
        self.num_latent = num_latent
        # gpflow.models.SVGP._init_variational_parameters
        self.q_mu = torch.nn.Parameter(torch.zeros((m, num_latent), dtype=torch.float64, device=dev))
        if q_diag:
            self.q_sqrt = torch.nn.Parameter(torch.ones((m, num_latent), dtype=torch.float64, device=dev))
        else:
            self.q_sqrt = torch.nn.Parameter(torch.eye(m, dtype=torch.float64, device=dev)[None].repeat(num_latent, 1, 1))

    def feature(self):
        """The feature object the dispatch functions of ``inducing_variables`` expect, viewing the current parameters."""
        if self._feat_cls is iv.InducingTensors:
            f = iv.InducingTensors(self.Z, self._num_levels, increments=self._increments)
        else:
            f = iv.InducingSequences(self.Z, self._num_levels)
        f.learn_weights = self._learn_weights
        if self._learn_weights:
            f.W = self.W
        return f

    def _q_sqrt(self):
        return self.q_sqrt if self.q_diag else torch.tril(self.q_sqrt)                               # models.py:48/:66

    def predict_f(self, X_new, full_cov=False, return_Kzz=False):
        """models.py:62-73."""
        Kzz, Kzx, Kxx = iv.Kuu_Kuf_Kff(self.feature(), self.kernel, X_new, jitter=JITTER, full_f_cov=full_cov)   # :65
        f_mean, f_var = base_conditional(Kzx, Kzz, Kxx, self.q_mu, full_cov=full_cov, q_sqrt=self._q_sqrt(), white=self.whiten)
        if self.mean_function is not None:
            f_mean = f_mean + self.mean_function(X_new)
        return (f_mean, f_var, Kzz) if return_Kzz else (f_mean, f_var)

    def elbo(self, X, Y):
        """models.py:40-59: sum of variational expectations, scaled to the full data set, minus the prior KL."""
        if self.whiten:
            f_mean, f_var = self.predict_f(X)
            KL = gauss_kl(self.q_mu, self._q_sqrt())                                                 # :47-48
        else:
            f_mean, f_var, Kzz = self.predict_f(X, return_Kzz=True)
            KL = gauss_kl(self.q_mu, self._q_sqrt(), K=Kzz)                                          # :50-51
        var_exp = self.likelihood.variational_expectations(f_mean, f_var, Y)                         # :54
        scale = float(self.num_data or X.shape[0]) / float(X.shape[0])                               # :57
        return var_exp.sum() * scale - KL

    def predict_y(self, X_new):
        f_mean, f_var = self.predict_f(X_new)
        return self.likelihood.predict_mean_and_var(f_mean, f_var)

    def fit(self, X, Y, iterations=100, lr=1e-2, minibatch_size=None, seed=0, callback=None, graph=False):
        """Maximise the ELBO with Adam.  Returns the ELBO trace.

        graph=True: after three eager iterations the whole step -- covariances (HIP kernels), conditional / KL / likelihood,
        backward pass (HIP gradient kernels + autograd) and the Adam update -- is recorded once as ONE HIP graph
        (torch.cuda.graph on a side stream; the library's calls are capture-safe once their scratch buffers and task lists
        exist) and replayed per iteration with the minibatch copied into static tensors: ~300 kernel launches per step become one
        graph launch (ts_classification notebook shape on MI355X: 338 -> 472 it/s kernel fixed, 287 -> 423 it/s kernel trainable,
        identical ELBO traces; profiles/r02_bench_grad.txt).  Shapes must not change between iterations (fixed minibatch size),
        and base kernels with a trainable parameter of their own (poly, mix) are not recorded: their value travels through
        the host per call."""
        gen = torch.Generator(device="cpu").manual_seed(seed)
        n = X.shape[0]
        if self.num_data is None:
            self.num_data = n
        mb = minibatch_size if (minibatch_size is not None and minibatch_size < n) else None

        def batch():
            if mb is None:
                return X, Y
            idx = torch.randperm(n, generator=gen)[:mb].to(X.device)
            return X[idx], Y[idx]

        params = [p for p in self.parameters() if p.requires_grad]
        trace = []
        if not graph:
            opt = torch.optim.Adam(params, lr=lr)
            for it in range(iterations):
                xb, yb = batch()
                opt.zero_grad()
                loss = -self.elbo(xb, yb)
                loss.backward()
                opt.step()
                trace.append(-loss.item())
                if callback is not None:
                    callback(it, trace[-1])
            return trace
        if getattr(self.kernel, "_has_p0", False):
            raise NotImplementedError("fit(graph=True): the base kernel's own parameter is handed to the library through the host per call")
        opt = torch.optim.Adam(params, lr=lr, capturable=True)
        xb, yb = batch()
        xs, ys = xb.clone(), yb.clone()
        dev = xs.device
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        warm = min(3, iterations)
        with torch.cuda.stream(side):
            for it in range(warm):          # eager, on the stream that will be captured: scratch buffers, task lists, Adam state
                if it > 0:
                    xb, yb = batch()
                    xs.copy_(xb); ys.copy_(yb)
                opt.zero_grad(set_to_none=True)
                loss = -self.elbo(xs, ys)
                loss.backward()
                opt.step()
                trace.append(-loss.item())
                if callback is not None:
                    callback(it, trace[-1])
        torch.cuda.current_stream(dev).wait_stream(side)
        if iterations <= warm:
            return trace
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g, stream=side):
            static_loss = -self.elbo(xs, ys)
            static_loss.backward()
            opt.step()
        # Replays cannot raise from inside (the recorded Cholesky factorisations are the unchecked form): the losses go into one
        # preallocated buffer, and a loss that stopped being finite -- a covariance that lost positive definiteness -- restores the
        # last parameters seen with a finite loss and raises, instead of letting Adam run on NaNs.
        trace_dev = torch.empty(iterations - warm, dtype=static_loss.dtype, device=dev)
        check_every = 25
        good = [p_.detach().clone() for p_ in params]
        good_opt = copy.deepcopy(opt.state_dict())
        for it in range(warm, iterations):
            xb, yb = batch()
            xs.copy_(xb); ys.copy_(yb)
            g.replay()
            trace_dev[it - warm].copy_(static_loss.detach())
            if callback is not None:
                callback(it, -float(static_loss.item()))
            if (it - warm) % check_every == check_every - 1 or it == iterations - 1:
                lo = (it - warm) // check_every * check_every
                # (a recorded loss is the one of the parameters BEFORE that replay's update: finite losses alone do not say that the last
                # update -- finite loss, infinite gradient -- left finite parameters, so those are looked at as well)
                finite = bool(torch.isfinite(trace_dev[lo:it - warm + 1]).all()) and all(bool(torch.isfinite(p_).all()) for p_ in params)
                if not finite:
                    with torch.no_grad():
                        for p_, g_ in zip(params, good):
                            p_.copy_(g_)
                    opt.load_state_dict(good_opt)        # Adam's moments saw the same NaNs
                    raise FloatingPointError("fit(graph=True): the ELBO stopped being finite between iterations %d and %d "
                                             "(a covariance matrix lost positive definiteness?); parameters and optimiser state "
                                             "restored to iteration %d" % (warm + lo, it, warm + lo))
                good = [p_.detach().clone() for p_ in params]
                good_opt = copy.deepcopy(opt.state_dict())
        trace.extend((-trace_dev).tolist())
        self._step_graph = g             # keeps the recorded step (and the memory it owns) alive with the model
        return [float(t) for t in trace]
