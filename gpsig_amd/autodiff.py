"""Differentiable signature kernels: the training-time counterpart of ``gpsig_amd.kernels``.

The reference is trained by TensorFlow's autodiff straight through ``SignatureKernel.K*`` (gpsig/training.py:149-164,
gpsig/models.py:40-59).  Here the same role is split in two:

* the heavy part -- the level recursions ``_K_seq``, ``_K_seq_diag``, ``_K_tens``, ``_K_tens_vs_seq`` (gpsig/kernels.py:188-340)
  and their reverse-mode derivatives -- runs in the HIP library (forward: the fused kernels of ``gpsig_*_levels``; backward:
  ``gpsig_*_levels_grad``), wrapped as ``torch.autograd.Function``s;
* the light, elementwise part -- lengthscale / lag scaling (kernels.py:343-398), level normalisation, ``sigma * variances``
  and the level sum (kernels.py:401-671) -- is written with torch ops on the GPU so that autograd chains it.

``SignatureKernelModule`` holds the hyper-parameters as unconstrained ``torch.nn.Parameter``s with GPflow 1.5.1's transforms
(``transforms.positive`` = softplus + 1e-6 for variances, sigma, lengthscales, gamma, the base-kernel parameter;
``transforms.Logistic`` for lags; kernels.py:65-88) so that an optimiser step means what it means in the reference.
Exact mode: first- and higher-order algorithms through those kernels.  Low-rank mode (kernels.py:239-311, :424-426, :442-458; a training
option of the reference's benchmark driver, benchmarks/models/train_gpsig.py:21, :58): the Nystrom features, their whitening (an eigendecomposition that
autograd differentiates, as TensorFlow does: low_rank_calculations.py:50-60), the running sums and the sparse projections are torch
ops on the GPU (GEMMs, gathers, rocSOLVER), with the landmarks GATHERED from the scaled inputs so that gradients reach them too; sized
for training batches (a projection materialises (N, L, non-zeros) products).  Computed by the float64 kernels (float32 tensors -- a module after
.float() -- are converted on the way in, results and gradients rounded on the way out).  No CPU fallback: tensors must live on the GPU.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .kernels import JITTER, SignatureKernel

_POS_LOWER = 1e-6   # gpflow.transforms.positive = Log1pe(lower=1e-6)


def positive(raw):
    """gpflow.transforms.Log1pe.forward: softplus(x) + 1e-6."""
    return torch.nn.functional.softplus(raw) + _POS_LOWER


def positive_inverse(value):
    """gpflow.transforms.Log1pe.backward."""
    y = np.asarray(value, dtype=np.float64) - _POS_LOWER
    return np.where(y > 35.0, y, np.log(np.expm1(np.maximum(y, 1e-300))))


def logistic_inverse(value):
    v = np.asarray(value, dtype=np.float64)
    return np.log(v) - np.log1p(-v)


class _Spec:
    """What a level primitive needs besides its tensors."""

    def __init__(self, base, num_levels, difference, p1=0.0, order=1):
        self.base, self.num_levels, self.difference, self.p1 = base, int(num_levels), bool(difference), float(p1)
        self.order = max(1, min(int(order), self.num_levels))          # kernels.py:57 clamps to num_levels

    def params(self, d_cols, p0, keep):
        p = _lib.Params()
        p.base_kernel = _lib.BASE[self.base]
        p.dtype = _lib.F64
        p.num_features, p.num_levels, p.order = int(d_cols), self.num_levels, self.order
        p.difference, p.normalization, p.num_lags = int(self.difference), 0, 0
        p.sigma, p.jitter = 1.0, JITTER
        p.base_params[0], p.base_params[1] = float(p0), self.p1
        ones = np.ones(self.num_levels + 1)
        keep.append(ones)
        p.variances = ones.ctypes.data_as(C.POINTER(C.c_double))
        p.lengthscales = None
        return p


# The library's "wide" option (state spaces beyond the exact-shape kernels' columns: kernel arguments by dgemm + fused map / difference / recursion
# kernels, csrc/wide_api.hip): None = where the exact-shape kernels are not built (the library's default), True = wherever built, False = never.
_WIDE = {"value": -1}
WIDE_BASES = ("rbf", "matern12", "matern32", "matern52")
WIDE_PRIMITIVES = {"tvs", "diag", "seq", "tens"}           # the level primitives the library has a wide route for
WIDE_LAT_MAX_COLS = 512                            # ... the sequence lattices up to this many columns (csrc/wide_api.hip)


def set_wide_route(mode):
    _WIDE["value"] = {None: -1, True: 1, False: 0}[mode]


def _ctx_for(t):
    if not t.is_cuda:
        raise RuntimeError("gpsig_amd.autodiff needs CUDA (ROCm) tensors: there is no CPU path")
    ctx = _lib.context(t.device.index or 0, torch.cuda.current_stream(t.device).cuda_stream)
    ctx.set_pointer_mode(_lib.PTR_DEVICE)
    if getattr(ctx, "_wide_option", -1) != _WIDE["value"]:
        ctx.set_option("wide", _WIDE["value"])
        ctx._wide_option = _WIDE["value"]
    return ctx


def _c(t):
    return t.detach().to(torch.float64).contiguous()


def _out_dtype(*ts):
    """float32 only if every tensor argument is float32 (then the float64 kernels' result is rounded, as the evaluation
    path does for float32 requests it has no float32 kernel for); anything else is float64."""
    ts = [t for t in ts if t is not None]
    return torch.float32 if ts and all(t.dtype == torch.float32 for t in ts) else torch.float64


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _p0_value(p0):
    return float(p0.detach().cpu()) if p0 is not None else 0.0


class _SeqGramLevels(torch.autograd.Function):
    """_K_seq (kernels.py:208-237): scaled sequences (N1, L1, d) [, (N2, L2, d)] -> (M+1, N1, N2).

    A forward pass that will be differentiated asks the library to keep what its reverse pass needs of the forward recursion
    (gpsig_seq_gram_levels_stash: where the fused reverse kernel can continue from it, the backward call then runs the backward sweep only);
    the library says whether it did, and whether the stash still stands when the backward call comes."""

    @staticmethod
    def forward(ctx, Xs, X2s, p0, spec):
        X, X2 = _c(Xs), (None if X2s is None else _c(X2s))
        n1, l1, d = X.shape
        n2, l2 = (n1, l1) if X2 is None else X2.shape[:2]
        keep = []
        p = spec.params(d, _p0_value(p0), keep)
        out = torch.empty((spec.num_levels + 1, n1, n2), dtype=torch.float64, device=X.device)
        ctx.stash = None
        want = (X.is_cuda and spec.base in ("rbf", "matern12", "matern32", "matern52")
                and (Xs.requires_grad or (X2s is not None and X2s.requires_grad)))
        if want:
            desc = (C.c_int64 * 8)()
            lib_ctx = _ctx_for(X)
            lib_ctx.call("gpsig_seq_gram_levels_stash", p, _ptr(X), None if X2 is None else _ptr(X2), n1, n2, l1, l2, _ptr(out), desc)
            if desc[0] != 0:
                ctx.stash = list(desc)
                ctx.stash_ctx = lib_ctx            # the stash lives in THIS context: a backward pass on another stream must not look for it elsewhere
        else:
            _ctx_for(X).call("gpsig_seq_gram_levels", p, _ptr(X), None if X2 is None else _ptr(X2), n1, n2, l1, l2, _ptr(out))
        ctx.spec, ctx.has_x2, ctx.has_p0 = spec, X2 is not None, p0 is not None
        ctx.dt = (Xs.dtype, None if X2s is None else X2s.dtype)
        ctx.save_for_backward(X, X2 if X2 is not None else X.new_empty(0), p0 if p0 is not None else X.new_empty(0))
        return out.to(_out_dtype(Xs, X2s))

    @staticmethod
    def backward(ctx, G):
        X, X2, p0 = ctx.saved_tensors
        X2 = X2 if ctx.has_x2 else None
        n1, l1, d = X.shape
        n2, l2 = (n1, l1) if X2 is None else X2.shape[:2]
        keep = []
        p = ctx.spec.params(d, _p0_value(p0) if ctx.has_p0 else 0.0, keep)
        G = _c(G)
        gX = torch.empty_like(X)
        gX2 = None if X2 is None else torch.empty_like(X2)
        gb = torch.zeros(2, dtype=torch.float64, device=X.device)
        taken = C.c_int32(0)
        if ctx.stash is not None:
            desc = (C.c_int64 * 8)(*ctx.stash)
            if ctx.stash_ctx is _ctx_for(X):       # same (device, stream) as the forward pass: its stream order makes the stash valid here
                ctx.stash_ctx.call("gpsig_seq_gram_levels_grad_stash", p, _ptr(X), None if X2 is None else _ptr(X2), n1, n2, l1, l2, _ptr(G), _ptr(gX),
                                   None if gX2 is None else _ptr(gX2), desc, C.byref(taken))
        if not taken.value:
            _ctx_for(X).call("gpsig_seq_gram_levels_grad", p, _ptr(X), None if X2 is None else _ptr(X2), n1, n2, l1, l2, _ptr(G), _ptr(gX),
                             None if gX2 is None else _ptr(gX2), C.cast(gb.data_ptr(), C.POINTER(C.c_double)))
        gp0 = gb[0].to(p0.device).reshape(p0.shape).to(p0.dtype) if ctx.has_p0 else None
        return gX.to(ctx.dt[0]), None if gX2 is None else gX2.to(ctx.dt[1]), gp0, None


class _SeqGramSum(torch.autograd.Function):
    """K = sum_m w_m K_m [normalised] (kernels.py:401-476 after the scaling) of SignatureLinear / SignatureCosine as ONE op: forward
    gpsig_kernel_K (the evaluation path's feature contraction), backward gpsig_kernel_K_grad -- one product of the upstream with
    the features of every level at once, so the (M+1, N1, N2) level arrays and their gradients never exist."""

    @staticmethod
    def weights_on_host(w):
        return np.ascontiguousarray(w.detach().to(torch.float64).cpu().numpy())

    @staticmethod
    def _params(spec, d, wh, normalization, keep):
        p = spec.params(d, 0.0, keep)
        keep.append(wh)
        p.variances = wh.ctypes.data_as(C.POINTER(C.c_double))
        p.sigma, p.normalization = 1.0, int(bool(normalization))
        return p

    @staticmethod
    def applies(Xs, X2s, spec, normalization, wh=None):
        """Whether the library takes this shape through the feature space (asked before the forward pass commits to the route; the
        answer does not depend on the weights' values, so none are fetched from the device for it)."""
        if spec.base not in ("linear", "cosine") or not Xs.is_cuda:
            return False
        if wh is None:
            wh = np.ones(spec.num_levels + 1)
        n1, l1, d = Xs.shape                                    # (shapes only: the pointers say which arguments are there, nothing is read)
        n2, l2 = (n1, l1) if X2s is None else X2s.shape[:2]
        keep = []
        p = _SeqGramSum._params(spec, d, wh, normalization, keep)
        taken = C.c_int32(0)
        _ctx_for(Xs).call("gpsig_kernel_K_grad", p, _ptr(Xs), None if X2s is None else _ptr(X2s), n1, n2, l1, l2, None, None, None, None, C.byref(taken))
        return bool(taken.value)

    @staticmethod
    def forward(ctx, Xs, X2s, w, wh, spec, normalization):
        X, X2 = _c(Xs), (None if X2s is None else _c(X2s))
        n1, l1, d = X.shape
        n2, l2 = (n1, l1) if X2 is None else X2.shape[:2]
        keep = []
        p = _SeqGramSum._params(spec, d, wh, normalization, keep)
        ctx.wh = wh
        out = torch.empty((n1, n2), dtype=torch.float64, device=X.device)
        _ctx_for(X).call("gpsig_kernel_K", p, _ptr(X), None if X2 is None else _ptr(X2), n1, n2, l1, l2, 0, _ptr(out))
        ctx.spec, ctx.has_x2, ctx.normalization = spec, X2 is not None, bool(normalization)
        ctx.dt = (Xs.dtype, None if X2s is None else X2s.dtype)
        ctx.save_for_backward(X, X2 if X2 is not None else X.new_empty(0), w)
        return out.to(_out_dtype(Xs, X2s))

    @staticmethod
    def backward(ctx, g):
        X, X2, w = ctx.saved_tensors
        X2 = X2 if ctx.has_x2 else None
        n1, l1, d = X.shape
        n2, l2 = (n1, l1) if X2 is None else X2.shape[:2]
        keep = []
        p = _SeqGramSum._params(ctx.spec, d, ctx.wh, ctx.normalization, keep)
        g = _c(g)
        gX = torch.empty_like(X)
        gX2 = None if X2 is None else torch.empty_like(X2)
        gw = torch.zeros(ctx.spec.num_levels + 1, dtype=torch.float64, device=X.device)
        taken = C.c_int32(0)
        _ctx_for(X).call("gpsig_kernel_K_grad", p, _ptr(X), None if X2 is None else _ptr(X2), n1, n2, l1, l2, _ptr(g), _ptr(gX),
                         None if gX2 is None else _ptr(gX2), _ptr(gw), C.byref(taken))
        if not taken.value:
            raise RuntimeError("gpsig_kernel_K_grad declined a call its probe had accepted")
        # level 0 is constant (1; normalised: 1 / (1 + jitter) off the diagonal), a normalised symmetric Gram's diagonal is sum_m w_m
        tot = g.sum()
        if ctx.normalization and X2 is None:
            tr = torch.diagonal(g).sum()
            gw[0] = (tot - tr) / (1.0 + JITTER) + tr
            gw[1:] += tr
        else:
            gw[0] = tot / (1.0 + JITTER) if ctx.normalization else tot
        return gX.to(ctx.dt[0]), None if gX2 is None else gX2.to(ctx.dt[1]), gw.to(w.dtype), None, None, None


class _SigFeatures(torch.autograd.Function):
    """The explicit level features Phi(x) of the linear (cosine: of the unit vectors) kernel: scaled sequences (N, L, d) -> (N, ld), columns
    [0, F) = levels 1..M in the natural order of their multi-indices (gpsig_seq_features; backward: the reverse feature sweep,
    gpsig_seq_features_grad).  K_m(x, y) = <Phi_m(x), Phi_m(y)> (signature_algs.py:8-74), K_m(z, x) = <z_1 (x) .. (x) z_m, Phi_m(x)> (:101-160)."""

    @staticmethod
    def ld(spec, d, L):
        """Row stride of the features, or 0 where the feature kernels are not built (then the callers use the recursions)."""
        if spec.base not in ("linear", "cosine"):
            return 0
        keep = []
        return int(_lib.load().gpsig_seq_features_ld(C.byref(spec.params(d, 0.0, keep)), int(L)))

    @staticmethod
    def forward(ctx, Xs, spec):
        X = _c(Xs)
        n, l, d = X.shape
        keep = []
        p = spec.params(d, 0.0, keep)
        ld = int(_lib.load().gpsig_seq_features_ld(C.byref(p), l))
        out = torch.empty((n, ld), dtype=torch.float64, device=X.device)
        _ctx_for(X).call("gpsig_seq_features", p, _ptr(X), n, l, _ptr(out))
        ctx.spec, ctx.dt = spec, Xs.dtype
        ctx.save_for_backward(X, out)
        return out

    @staticmethod
    def backward(ctx, G):
        X, Phi = ctx.saved_tensors
        n, l, d = X.shape
        keep = []
        p = ctx.spec.params(d, 0.0, keep)
        G = _c(G)
        gX = torch.empty_like(X)
        _ctx_for(X).call("gpsig_seq_features_grad", p, _ptr(X), n, l, _ptr(Phi), _ptr(G), _ptr(gX))
        return gX.to(ctx.dt), None


def signature_features(X, num_levels, order=1, difference=True, unit_points=False):
    """Explicit signature-level features of a batch of sequences X (N, L, d) on the GPU, differentiable: a list of M tensors, level m of
    shape (N, d^m) (views into one buffer), with <Phi_m(x), Phi_m(y)> = level m of SignatureLinear(order=order, difference=difference)
    (unit_points: of SignatureCosine).  order = num_levels and difference = True: the signature of the piecewise-linear path through the
    points, truncated at num_levels -- what the reference's notebook compares its kernel against (esig)."""
    spec = _Spec("cosine" if unit_points else "linear", num_levels, difference, order=order)
    n, l, d = X.shape
    if not _SigFeatures.ld(spec, d, l):
        raise NotImplementedError("signature_features: 2 <= num_levels <= 8, d <= 32 and a sequence's arrays within the LDS")
    return _split_levels(_SigFeatures.apply(X, spec), d, num_levels)


def _split_levels(Phi, d, M):
    out, off = [], 0
    for m in range(1, M + 1):
        out.append(Phi[:, off:off + d ** m])
        off += d ** m
    return out


_COL_LEVELS = {}


def _col_levels(d, M, ld, device):
    """Level of every column of a feature buffer (N, ld): 1..M for the levels' columns, 0 for the level-0 column, M+1 for the padding."""
    key = (d, M, ld, str(device))
    if key not in _COL_LEVELS:
        idx = np.full(ld, M + 1, dtype=np.int64)
        off = 0
        for m in range(1, M + 1):
            idx[off:off + d ** m] = m
            off += d ** m
        idx[off] = 0
        _COL_LEVELS[key] = torch.as_tensor(idx, device=device)
    return _COL_LEVELS[key]


class _ScaleLevels(torch.autograd.Function):
    """Phi (N, ld) with level m's columns multiplied by fac[m][n] (fac: (M+1, N); padding columns stay zero): the per-sequence factors of
    kernels.py:576-584 / :656-667 put onto the features, whole-buffer operations instead of one autograd slice per level."""

    @staticmethod
    def forward(ctx, Phi, fac, d):
        M = fac.shape[0] - 1
        col = _col_levels(d, M, Phi.shape[1], Phi.device)
        facx = torch.cat([fac, fac.new_zeros(1, fac.shape[1])], dim=0)                              # level M+1: the padding
        colfac = facx.T[:, col]                                                                     # (N, ld)
        ctx.save_for_backward(Phi, colfac)
        ctx.d, ctx.M = d, M
        return Phi * colfac

    @staticmethod
    def backward(ctx, G):
        Phi, colfac = ctx.saved_tensors
        d, M = ctx.d, ctx.M
        G = G.contiguous()
        gp = G * Phi
        rows, off = [None] * (M + 1), 0
        for m in range(1, M + 1):
            rows[m] = gp[:, off:off + d ** m].sum(dim=1)
            off += d ** m
        rows[0] = gp[:, off]
        return G * colfac, torch.stack(rows, dim=0), None


class _LevelNorms(torch.autograd.Function):
    """K_m(x, x) = |Phi_m(x)|^2 of every level: (N, ld) -> (M+1, N) (level 0: 1)."""

    @staticmethod
    def forward(ctx, Phi, d, M):
        sq = Phi * Phi
        rows, off = [None] * (M + 1), 0
        for m in range(1, M + 1):
            rows[m] = sq[:, off:off + d ** m].sum(dim=1)
            off += d ** m
        rows[0] = sq[:, off]
        ctx.save_for_backward(Phi)
        ctx.d, ctx.M = d, M
        return torch.stack(rows, dim=0)

    @staticmethod
    def backward(ctx, G):
        (Phi,) = ctx.saved_tensors
        col = _col_levels(ctx.d, ctx.M, Phi.shape[1], Phi.device)
        Gx = torch.cat([G, G.new_zeros(1, G.shape[1])], dim=0)
        Gx = Gx.clone()
        Gx[0] = 0.0                                                                                 # level 0 is the constant 1
        return 2.0 * Phi * Gx.T[:, col], None, None


class _FeatureProduct(torch.autograd.Function):
    """A (T, F) @ B (N, F)^T with T << N: the backward's dA = G B has few result tiles and depth N, which the library's GEMM selection runs at a
    quarter of the speed of the same flops as a batch of products over chunks of N (1.78 -> 0.51 ms at T = 512, N = 16,384, F = 1,568)."""

    @staticmethod
    def forward(ctx, A, B):
        ctx.save_for_backward(A, B)
        return A @ B.T

    @staticmethod
    def backward(ctx, G):
        A, B = ctx.saved_tensors
        G = G.contiguous()              # (the gradient of a plain .sum() arrives as a stride-0 expansion, which the batched product runs 40 times slower on)
        T, N = G.shape
        ch = next((c for c in (32, 16, 8, 4, 2) if N % c == 0 and N // c >= 256), 1)
        if ch > 1:
            dA = torch.bmm(G.reshape(T, ch, N // ch).permute(1, 0, 2), B.reshape(ch, N // ch, B.shape[1])).sum(dim=0)
        else:
            dA = G @ B
        return dA, G.T @ A


def _tensor_features(Zs, M, increments, unit):
    """Rank-one inducing tensors as level features: Z (lt, T, d) [(lt, T, 2, d): increments, kernels.py:329-330] -> [(T, d^m)], level m the outer
    product of its m components, first index = the component paired with the earliest time (signature_algs.py:118-125)."""
    if unit:                                                                                        # kernels.py:820-828 on the tensors' side
        Zs = Zs / torch.sqrt(torch.square(Zs).sum(dim=-1, keepdim=True))
    if increments:
        Zs = Zs[:, :, 1] - Zs[:, :, 0]
    out, k = [], 0
    for m in range(1, M + 1):
        f = Zs[k]
        for j in range(1, m):
            f = (f[:, :, None] * Zs[k + j][:, None, :]).reshape(f.shape[0], -1)
        out.append(f)
        k += m
    return out


class _SeqDiagLevels(torch.autograd.Function):
    """_K_seq_diag (kernels.py:188-205): (N, L, d) -> (M+1, N)."""

    @staticmethod
    def forward(ctx, Xs, p0, spec):
        X = _c(Xs)
        n, l, d = X.shape
        keep = []
        p = spec.params(d, _p0_value(p0), keep)
        out = torch.empty((spec.num_levels + 1, n), dtype=torch.float64, device=X.device)
        _ctx_for(X).call("gpsig_seq_diag_levels", p, _ptr(X), n, l, _ptr(out))
        ctx.spec, ctx.has_p0 = spec, p0 is not None
        ctx.dt = Xs.dtype
        ctx.save_for_backward(X, p0 if p0 is not None else X.new_empty(0))
        return out.to(_out_dtype(Xs))

    @staticmethod
    def backward(ctx, G):
        X, p0 = ctx.saved_tensors
        n, l, d = X.shape
        keep = []
        p = ctx.spec.params(d, _p0_value(p0) if ctx.has_p0 else 0.0, keep)
        G = _c(G)
        gX = torch.empty_like(X)
        gb = torch.zeros(2, dtype=torch.float64, device=X.device)
        _ctx_for(X).call("gpsig_seq_diag_levels_grad", p, _ptr(X), n, l, _ptr(G), _ptr(gX), C.cast(gb.data_ptr(), C.POINTER(C.c_double)))
        gp0 = gb[0].to(p0.device).reshape(p0.shape).to(p0.dtype) if ctx.has_p0 else None
        return gX.to(ctx.dt), gp0, None


class _TensGramLevels(torch.autograd.Function):
    """_K_tens (kernels.py:263-283): scaled inducing tensors (lt, T, [2,] d) -> (M+1, T, T)."""

    @staticmethod
    def forward(ctx, Zs, p0, spec, increments):
        Z = _c(Zs)
        t, d = Z.shape[1], Z.shape[-1]
        keep = []
        p = spec.params(d, _p0_value(p0), keep)
        out = torch.empty((spec.num_levels + 1, t, t), dtype=torch.float64, device=Z.device)
        _ctx_for(Z).call("gpsig_tens_gram_levels", p, _ptr(Z), t, int(bool(increments)), _ptr(out))
        ctx.spec, ctx.has_p0, ctx.increments = spec, p0 is not None, bool(increments)
        ctx.dt = Zs.dtype
        ctx.save_for_backward(Z, p0 if p0 is not None else Z.new_empty(0))
        return out.to(_out_dtype(Zs))

    @staticmethod
    def backward(ctx, G):
        Z, p0 = ctx.saved_tensors
        t, d = Z.shape[1], Z.shape[-1]
        keep = []
        p = ctx.spec.params(d, _p0_value(p0) if ctx.has_p0 else 0.0, keep)
        G = _c(G)
        gZ = torch.empty_like(Z)
        gb = torch.zeros(2, dtype=torch.float64, device=Z.device)
        _ctx_for(Z).call("gpsig_tens_gram_levels_grad", p, _ptr(Z), t, int(ctx.increments), _ptr(G), _ptr(gZ),
                         C.cast(gb.data_ptr(), C.POINTER(C.c_double)))
        gp0 = gb[0].to(p0.device).reshape(p0.shape).to(p0.dtype) if ctx.has_p0 else None
        return gZ.to(ctx.dt), gp0, None, None


class _TensVsSeqLevels(torch.autograd.Function):
    """_K_tens_vs_seq (kernels.py:313-340): (lt, T, [2,] d), (N, L, d) -> (M+1, T, N)."""

    @staticmethod
    def forward(ctx, Zs, Xs, p0, spec, increments):
        Z, X = _c(Zs), _c(Xs)
        t, d = Z.shape[1], Z.shape[-1]
        n, l = X.shape[:2]
        keep = []
        p = spec.params(d, _p0_value(p0), keep)
        out = torch.empty((spec.num_levels + 1, t, n), dtype=torch.float64, device=Z.device)
        _ctx_for(Z).call("gpsig_tens_vs_seq_levels", p, _ptr(Z), _ptr(X), t, n, l, int(bool(increments)), _ptr(out))
        ctx.spec, ctx.has_p0, ctx.increments = spec, p0 is not None, bool(increments)
        ctx.dt = (Zs.dtype, Xs.dtype)
        ctx.save_for_backward(Z, X, p0 if p0 is not None else Z.new_empty(0))
        return out.to(_out_dtype(Zs, Xs))

    @staticmethod
    def backward(ctx, G):
        Z, X, p0 = ctx.saved_tensors
        t, d = Z.shape[1], Z.shape[-1]
        n, l = X.shape[:2]
        keep = []
        p = ctx.spec.params(d, _p0_value(p0) if ctx.has_p0 else 0.0, keep)
        G = _c(G)
        gZ, gX = torch.empty_like(Z), torch.empty_like(X)
        gb = torch.zeros(2, dtype=torch.float64, device=Z.device)
        _ctx_for(Z).call("gpsig_tens_vs_seq_levels_grad", p, _ptr(Z), _ptr(X), t, n, l, int(ctx.increments), _ptr(G), _ptr(gZ), _ptr(gX),
                         C.cast(gb.data_ptr(), C.POINTER(C.c_double)))
        gp0 = gb[0].to(p0.device).reshape(p0.shape).to(p0.dtype) if ctx.has_p0 else None
        return gZ.to(ctx.dt[0]), gX.to(ctx.dt[1]), gp0, None, None


class _TensVsSeqWeighted(torch.autograd.Function):
    """sum_m fac[m][n] * _K_tens_vs_seq(Z, X)[m][t][n] -> (T, N): the level sum of kernels.py:572-588 / :638-667 taken inside the HIP
    kernel (gpsig_tens_vs_seq_weighted), so that a training step never holds the (M+1, T, N) level array or its gradient; the
    per-sequence factors fac (M+1, N) -- sigma * variances / sqrt(level diagonals + jitter) -- are a differentiable input."""

    @staticmethod
    def forward(ctx, Zs, Xs, fac, p0, spec, increments):
        Z, X = _c(Zs), _c(Xs)
        F = _c(fac).t().contiguous()                                  # (N, M+1): what a lane reads per sequence
        t, d = Z.shape[1], Z.shape[-1]
        n, l = X.shape[:2]
        keep = []
        p = spec.params(d, _p0_value(p0), keep)
        out = torch.empty((t, n), dtype=torch.float64, device=Z.device)
        # when a reverse pass will follow: room for the chain totals of every (tensor, sequence) pair, which the tile kernel leaves
        # on its way and the reverse pass would otherwise rebuild with a forward sweep of its own
        aux, wrote = None, C.c_int32(0)
        if any(ctx.needs_input_grad[:4]):
            aux = torch.empty(int(_lib.load().gpsig_tens_vs_seq_aux_elems(C.byref(p), t, n)), dtype=torch.float64, device=Z.device)
        _ctx_for(Z).call("gpsig_tens_vs_seq_weighted", p, _ptr(Z), _ptr(X), t, n, l, int(bool(increments)), _ptr(F), _ptr(out),
                         None if aux is None else _ptr(aux), C.byref(wrote))
        if not wrote.value:
            aux = None
        ctx.spec, ctx.has_p0, ctx.increments, ctx.has_aux = spec, p0 is not None, bool(increments), aux is not None
        ctx.dt = (Zs.dtype, Xs.dtype, fac.dtype)
        ctx.save_for_backward(Z, X, F, p0 if p0 is not None else Z.new_empty(0), aux if aux is not None else Z.new_empty(0))
        return out.to(_out_dtype(Zs, Xs, fac))

    @staticmethod
    def backward(ctx, G):
        Z, X, F, p0, aux = ctx.saved_tensors
        t, d = Z.shape[1], Z.shape[-1]
        n, l = X.shape[:2]
        keep = []
        p = ctx.spec.params(d, _p0_value(p0) if ctx.has_p0 else 0.0, keep)
        G = _c(G)
        gZ, gX, gF = torch.empty_like(Z), torch.empty_like(X), torch.empty_like(F)
        gb = torch.zeros(2, dtype=torch.float64, device=Z.device)
        _ctx_for(Z).call("gpsig_tens_vs_seq_weighted_grad", p, _ptr(Z), _ptr(X), t, n, l, int(ctx.increments), _ptr(F), _ptr(G),
                         _ptr(aux) if ctx.has_aux else None, _ptr(gZ), _ptr(gX), _ptr(gF), C.cast(gb.data_ptr(), C.POINTER(C.c_double)))
        gp0 = gb[0].to(p0.device).reshape(p0.shape).to(p0.dtype) if ctx.has_p0 else None
        return gZ.to(ctx.dt[0]), gX.to(ctx.dt[1]), gF.t().to(ctx.dt[2]), gp0, None, None


# ---- the "matrix route": base-kernel tensors built here, the recursions on them in the library --------------------------------------
# For state spaces beyond the gradient kernels' 64 columns and for SignatureSpectral (whose alpha, omega, gamma are trained:
# gpsig/kernels.py:912-914) the base-kernel tensor of kernels.py:188-340 is built with torch ops on the GPU -- at that width it is a
# d-deep contraction, a rocBLAS GEMM (north_star: "MFMA only where it is a true contraction") -- and differentiated by autograd;
# the library does what gpsig/signature_algs.py does with that tensor: the recursions on its increment lattices and their reverse
# passes (gpsig_lattice_levels / gpsig_chain_levels and their _grad).  The lattices are held in memory: minibatch-sized problems.
class _LatticeLevels(torch.autograd.Function):
    """signature_kern_first_order / _higher_order (signature_algs.py:28-35, :58-74) on increment lattices dM (P, R1, R2) -> (M+1, P)."""

    @staticmethod
    def forward(ctx, dM, spec):
        A = _c(dM)
        P, R1, R2 = A.shape
        keep = []
        p = spec.params(1, 0.0, keep)
        out = torch.empty((spec.num_levels + 1, P), dtype=torch.float64, device=A.device)
        _ctx_for(A).call("gpsig_lattice_levels", p, _ptr(A), P, R1, R2, _ptr(out))
        ctx.spec, ctx.dt = spec, dM.dtype
        ctx.save_for_backward(A)
        return out.to(_out_dtype(dM))

    @staticmethod
    def backward(ctx, G):
        (A,) = ctx.saved_tensors
        P, R1, R2 = A.shape
        keep = []
        p = ctx.spec.params(1, 0.0, keep)
        gA = torch.empty_like(A)
        _ctx_for(A).call("gpsig_lattice_levels_grad", p, _ptr(A), P, R1, R2, _ptr(_c(G)), _ptr(gA))
        return gA.to(ctx.dt), None


class _ChainLevels(torch.autograd.Function):
    """signature_kern_tens_vs_seq_first_order (signature_algs.py:116-127) on component increments m (lt, R, P) -> (M+1, P)."""

    @staticmethod
    def forward(ctx, m, spec):
        A = _c(m)
        lt, R, P = A.shape
        keep = []
        p = spec.params(1, 0.0, keep)
        out = torch.empty((spec.num_levels + 1, P), dtype=torch.float64, device=A.device)
        _ctx_for(A).call("gpsig_chain_levels", p, _ptr(A), P, R, _ptr(out))
        ctx.spec, ctx.dt = spec, m.dtype
        ctx.save_for_backward(A)
        return out.to(_out_dtype(m))

    @staticmethod
    def backward(ctx, G):
        (A,) = ctx.saved_tensors
        lt, R, P = A.shape
        keep = []
        p = ctx.spec.params(1, 0.0, keep)
        gA = torch.empty_like(A)
        _ctx_for(A).call("gpsig_chain_levels_grad", p, _ptr(A), P, R, _ptr(_c(G)), _ptr(gA))
        return gA.to(ctx.dt), None


class _SqrtZeroGrad(torch.autograd.Function):
    """sqrt with derivative 0 at 0 (TensorFlow's, and torch's, is inf there, and 0 * inf = NaN reaches every parameter through the
    zero distances of a symmetric base-kernel tensor: the reference's 'exp' spectral family cannot be trained on K(X, X) as written,
    gpsig/kernels.py:924; the clamp of its own Matern kernels, :779-781, has the same effect as this)."""

    @staticmethod
    def forward(ctx, x):
        y = torch.sqrt(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return torch.where(y > 0, g / (2 * y), torch.zeros_like(g))


def base_kernel_matrix(base, A, B, p0=None, p1=0.0, spectral=None):
    """kappa(a_i, b_j) of gpsig/kernels.py:765-781, :799-993 for A (..., P, d), B (..., Q, d) -> (..., P, Q), torch ops (autograd)."""
    if base == "spectral":                                                                          # :921-942
        family, alpha, omega, gamma = spectral
        Q = alpha.shape[0]
        out = 0.0
        # on the difference tensor, as the reference (:923-925): the exponential envelope takes a square root of the squared distance,
        # and a distance assembled from norms and an inner product is rounding noise -- 1e-8 after the root -- where it should be zero
        diff = A[..., :, None, :] - B[..., None, :, :]
        for q in range(Q):
            sq = torch.square(diff * gamma[q]).sum(-1)
            gauss = family == "rbf" or (family == "mixed" and q < Q // 2)
            env = torch.exp(-sq / 2) if gauss else torch.exp(-_SqrtZeroGrad.apply(sq) / 2)          # :928 / :926
            out = out + alpha[q] * env * torch.cos(2.0 * math.pi * (diff * omega[q]).sum(-1))       # :937, :942
        return out
    inner = torch.matmul(A, B.transpose(-1, -2))
    if base == "linear":
        return inner
    As, Bs = (A * A).sum(-1), (B * B).sum(-1)
    if base == "cosine":
        return inner / (torch.sqrt(As)[..., :, None] * torch.sqrt(Bs)[..., None, :])
    if base == "poly":
        return (inner + p0) ** p1
    dist = -2 * inner + As[..., :, None] + Bs[..., None, :]
    if base == "rbf":
        return torch.exp(-dist / 2)
    if base == "mix":
        return p0 * torch.exp(-dist / 2) + (1.0 - p0) * inner
    r = torch.sqrt(torch.clamp(dist, min=1e-40))                                                    # :779-781
    if base == "matern12":
        return torch.exp(-r)
    if base == "matern32":
        return (1.0 + math.sqrt(3.0) * r) * torch.exp(-math.sqrt(3.0) * r)
    if base == "matern52":
        return (1.0 + math.sqrt(5.0) * r + 5.0 / 3.0 * r * r) * torch.exp(-math.sqrt(5.0) * r)
    raise ValueError(base)


# ---- scaling (gpsig/kernels.py:343-398, gpsig/lags.py) in torch ------------------------------------------------------
def _lin_interp(time, X, time_query):
    """gpsig/lags.py:7-38 (3-D branch :32-33).  X (N, L, d), time (L,), time_query (L, p) -> (N, L, p, d)."""
    dist = time[:, None, None] - time_query[None, :, :]                                             # :20
    masked = torch.where(dist > JITTER, torch.full_like(dist, -math.inf), dist)                     # :22
    left = torch.argmax(masked, dim=0)
    right = torch.clamp(left + 1, max=X.shape[1] - 1)                                               # :23
    Xl, Xr = X[:, left, :], X[:, right, :]                                                          # :25-26
    tl, tr = time[left], time[right]                                                                # :28-29
    return Xl + (time_query[None, ..., None] - tl[None, ..., None]) * (Xr - Xl) / (tr[None, ..., None] - tl[None, ..., None])  # :33


def _add_lags(X, lags):
    """gpsig/lags.py:41-63.  (N, L, d) -> (N, L, p+1, d)."""
    L = X.shape[1]
    time = torch.arange(L, dtype=X.dtype, device=X.device) / (L - 1)                                # :56
    time_lags = torch.clamp(time[:, None] - lags[None, :], min=0.)                                  # :57
    return torch.cat((X[:, :, None, :], _lin_interp(time, X, time_lags)), dim=2)                    # :59-61


class LowRankDraw:
    """The random objects of ONE low-rank evaluation that do not depend on values (the reference draws them inside the graph per
    evaluation: kernels.py:443-446, low_rank_calculations.py:12-20, :52, :92-101): which ``num_components`` of the evaluation's points are
    the Nystrom landmarks (indices into the concatenation the reference gathers from: [tensor points,] sequence points [, second
    sequences' points]), the jitter added to the landmark Gram, one sparse projection per level >= 2 (``gpsig_amd.low_rank.Sketch``)."""

    def __init__(self, idx, jitter_diag, sketches):
        self.idx = np.ascontiguousarray(idx, dtype=np.int64)
        self.jitter_diag = np.ascontiguousarray(jitter_diag, dtype=np.float64)
        self.sketches = list(sketches)
        self._dev = {}

    def on(self, device):
        """(idx, jitter_diag, [(i1, i2, val, column of every entry, r)]) as tensors on ``device``."""
        key = str(device)
        if key not in self._dev:
            t = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt, device=device)
            sk = []
            for s_ in self.sketches:
                col = np.repeat(np.arange(s_.r), np.diff(s_.colptr))
                i1, i2, val, colt = t(s_.i1, torch.int64), t(s_.i2, torch.int64), t(s_.val, torch.float64), t(col, torch.int64)
                # the projection as a dense (entries, r) matrix holding each entry's value in its output column: summing an entry's product
                # into its column is then a GEMM (index_add does it with atomics on r addresses per row: 6-14 ms for the inducing
                # tensors' six chained projections at the reference's default ranks; this: well under a millisecond)
                # ... and the two operand selections a[:, i1], b[:, i2] as products with 0 / 1 matrices (k1, entries), (k2, entries): the
                # reverse pass of a gather is a scatter-add with atomics as well, that of a product another product
                dense = None
                nnz = int(val.shape[0])
                if nnz * max(int(s_.r), int(s_.k1), int(s_.k2)) <= (1 << 24):
                    ar = torch.arange(nnz, device=device)
                    out_m = torch.zeros((nnz, int(s_.r)), dtype=torch.float64, device=device)
                    sel1 = torch.zeros((int(s_.k1), nnz), dtype=torch.float64, device=device)
                    sel2 = torch.zeros((int(s_.k2), nnz), dtype=torch.float64, device=device)
                    if nnz:
                        out_m[ar, colt] = val
                        sel1[i1, ar] = 1.0
                        sel2[i2, ar] = 1.0
                    dense = (sel1, sel2, out_m)
                sk.append((i1, i2, val, colt, int(s_.r), dense))
            self._dev[key] = (t(self.idx, torch.int64), t(self.jitter_diag, torch.float64), sk)
        return self._dev[key]


def _apply_sketch(sk, A, B):
    """lr_hadamard_prod_rand (low_rank_calculations.py:76-193) given its random matrix: out[..., j] = sum over the entries e of
    column j of val[e] A[..., i1[e]] B[..., i2[e]].  (..., k1), (..., k2) -> (..., r); rows in chunks of at most 2^27 products."""
    i1, i2, val, col, r, dense = sk
    lead = A.shape[:-1]
    A2, B2 = A.reshape(-1, A.shape[-1]), B.reshape(-1, B.shape[-1])
    rows, nnz = A2.shape[0], max(int(i1.shape[0]), 1)
    step = max(1, (1 << 27) // nnz)
    outs = []
    for r0 in range(0, rows, step):
        a, b = A2[r0:r0 + step], B2[r0:r0 + step]
        if dense is not None and dense[0].dtype == a.dtype:
            outs.append(((a @ dense[0]) * (b @ dense[1])) @ dense[2])
            continue
        prod = a[:, i1] * b[:, i2] * val
        outs.append(torch.zeros((a.shape[0], r), dtype=prod.dtype, device=prod.device).index_add(1, col, prod))
    out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
    return out.reshape(*lead, r)


def _sketch_array(sketches, keep):
    """gpsig_sketch[] of host-side ``low_rank.Sketch`` objects (the arrays stay alive in ``keep``)."""
    arr = (_lib.SketchC * max(len(sketches), 1))()
    for k, sk in enumerate(sketches):
        cp, i1, i2, val = (np.ascontiguousarray(sk.colptr, np.int32), np.ascontiguousarray(sk.i1, np.int32), np.ascontiguousarray(sk.i2, np.int32),
                           np.ascontiguousarray(sk.val, np.float64))
        keep.extend([cp, i1, i2, val])
        arr[k].k1, arr[k].k2, arr[k].r, arr[k].nnz = int(sk.k1), int(sk.k2), int(sk.r), int(val.shape[0])
        arr[k].colptr = cp.ctypes.data_as(C.POINTER(C.c_int32))
        arr[k].i1 = i1.ctypes.data_as(C.POINTER(C.c_int32))
        arr[k].i2 = i2.ctypes.data_as(C.POINTER(C.c_int32))
        arr[k].val = val.ctypes.data_as(C.POINTER(C.c_double))
    keep.append(arr)
    return arr


class _LrSeqFeatures(torch.autograd.Function):
    """_K_seq_lr_feat (kernels.py:239-261) given the landmarks and the whitening: scaled sequences (N, L, d), S (c, d), Wh (c, c) ->
    Phi (N, 1 + c + (M-1) r), by the fused HIP feature kernel; the reverse pass by lr_seq_features_grad_kernel (csrc/lr_grad_kernel.hpp):
    dPhi -> dX, dS, dWh, d base parameter.  Round 3 ran both directions as torch ops (_LowRankScope._seq_torch below: gather x gather x
    value + index_add over every (point, entry) product), 9.1 s for the SVGP covariances of BASELINE configs[2]."""

    @staticmethod
    def forward(ctx, Xs, S, Wh, p0, spec, sketches, r):
        X, Sd, Whd = _c(Xs), _c(S), _c(Wh)
        n, l, d = X.shape
        cc = Sd.shape[0]
        keep = []
        p = spec.params(d, _p0_value(p0), keep)
        arr = _sketch_array(sketches, keep)
        F = 1 + cc + (spec.num_levels - 1) * int(r)
        out = torch.empty((n, F), dtype=torch.float64, device=X.device)
        _ctx_for(X).call("gpsig_lr_seq_features_dev", p, cc, int(r), len(sketches), arr, _ptr(X), n, l, _ptr(Sd), _ptr(Whd), _ptr(out))
        ctx.spec, ctx.sketches, ctx.r, ctx.has_p0 = spec, sketches, int(r), p0 is not None
        ctx.dt = (Xs.dtype, S.dtype, Wh.dtype)
        ctx.save_for_backward(X, Sd, Whd, p0 if p0 is not None else X.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, G):
        X, Sd, Whd, p0 = ctx.saved_tensors
        n, l, d = X.shape
        cc = Sd.shape[0]
        keep = []
        p = ctx.spec.params(d, _p0_value(p0) if ctx.has_p0 else 0.0, keep)
        arr = _sketch_array(ctx.sketches, keep)
        G = _c(G)
        gX, gS, gWh = torch.empty_like(X), torch.empty_like(Sd), torch.empty_like(Whd)
        gb = torch.zeros(2, dtype=torch.float64, device=X.device)
        _ctx_for(X).call("gpsig_lr_seq_features_grad", p, cc, ctx.r, len(ctx.sketches), arr, _ptr(X), n, l, _ptr(Sd), _ptr(Whd), _ptr(G),
                         _ptr(gX), _ptr(gS), _ptr(gWh), C.cast(gb.data_ptr(), C.POINTER(C.c_double)))
        gp0 = gb[0].to(p0.device).reshape(p0.shape).to(p0.dtype) if ctx.has_p0 else None
        return gX.to(ctx.dt[0]), gS.to(ctx.dt[1]), gWh.to(ctx.dt[2]), gp0, None, None, None


class _LowRankScope:
    """One low-rank evaluation: landmarks gathered from the evaluation's scaled points, their whitening, and the level features of
    every input asked for (kept per tensor: Kzz, Kzx and Kxx of one call share them)."""

    def __init__(self, mod, pool, draw):
        idx, jd, self.sk = draw.on(pool.device)
        self.host_sketches = draw.sketches
        if int(idx.max()) >= pool.shape[0]:
            raise ValueError("the low-rank draw indexes %d points, the evaluation has %d" % (int(idx.max()) + 1, pool.shape[0]))
        self.mod = mod
        self.S = pool[idx]                                                                          # low_rank_calculations.py:47-48 (tf.gather: differentiable)
        W = mod._kappa(self.S, self.S) + torch.diag(jd)                                             # :51-52
        ev, U = torch.linalg.eigh(W)                                                                # :55
        # an eigenvector is fixed up to sign; the level >= 2 features depend on it through the projections of coordinate pairs: the
        # component of largest magnitude is taken positive (the library's and the checker's convention; constant under autograd)
        with torch.no_grad():
            top = U.abs().argmax(dim=0)
            sgn = torch.where(U[top, torch.arange(U.shape[1], device=U.device)] < 0, -1.0, 1.0).to(U.dtype)
        self.Wh = U * sgn[None, :] / torch.sqrt(ev + JITTER)[None, :]                               # :56-57, :60
        self._seq, self._tens = {}, {}

    def _nys(self, pts):
        return self.mod._kappa(pts, self.S) @ self.Wh                                               # :59-61

    def seq(self, Xs):
        """signature_algs.py:162-192 (with :191 summing P, as evidently intended).  (N, L, d') -> [(N, 1), (N, c), (N, r), ...].
        Through the HIP feature kernel and its reverse pass (_LrSeqFeatures) where they are built; torch ops otherwise (long sequences
        at large ranks, more than 64 components, module option ``lr_hip = False``)."""
        key = id(Xs)
        if key not in self._seq and getattr(self.mod, "lr_hip", True) and Xs.is_cuda:
            mod, kern = self.mod, self.mod.kern
            M, cc = kern.num_levels, int(self.S.shape[0])
            r = int(self.host_sketches[0].r) if self.host_sketches else int(kern.rank_bound)
            L, d = int(Xs.shape[1]), int(Xs.shape[2])
            # what csrc/lr_grad_api.hip takes: four (width, L) arrays of a sequence in LDS, at most 64 components
            rows, lp = max(cc, r, d, 16), (L + 63) // 64 * 64 + 1
            if kern._base != "spectral" and cc <= 64 and cc * d <= 4096 and 8 * lp * 4 * rows <= 156 * 1024 and M - 1 <= 7:
                try:
                    Phi = _LrSeqFeatures.apply(Xs, self.S, self.Wh, mod.p0, mod._spec, self.host_sketches, r)
                    cuts = [1, cc] + [r] * (M - 1)
                    self._seq[key] = (Xs, list(torch.split(Phi, cuts, dim=1)))
                except NotImplementedError:
                    pass
        if key not in self._seq:
            self._seq[key] = (Xs, self._seq_torch(Xs))
        return self._seq[key][1]

    def _seq_torch(self, Xs):
        """The same feature map as torch ops (round 3's route; kept as the checker of the HIP reverse pass and for shapes it is not
        built for)."""
        if True:
            N, L, d = Xs.shape
            U = self._nys(Xs.reshape(N * L, d)).reshape(N, L, -1)                                   # kernels.py:252-254
            if self.mod.kern.difference:
                U = U[:, 1:] - U[:, :-1]                                                            # :180
            Phi = [torch.ones((N, 1), dtype=U.dtype, device=U.device), U.sum(dim=1)]                # :177, :182
            P = U
            for i in range(2, self.mod.kern.num_levels + 1):
                P = torch.cumsum(P, dim=1) - P                                                      # :186 exclusive
                P = _apply_sketch(self.sk[i - 2], U, P)                                             # :188 / :190
                Phi.append(P.sum(dim=1))
            return Phi

    def tens(self, Zs, increments):
        """signature_algs.py:194-222, kernels.py:285-311.  (lt, T[, 2], d') -> [(T, 1), (T, c), (T, r), ...]."""
        key = id(Zs)
        if key not in self._tens:
            lt, T, d = Zs.shape[0], Zs.shape[1], Zs.shape[-1]
            if increments:
                F = self._nys(Zs.reshape(lt * T * 2, d)).reshape(lt, T, 2, -1)
                F = F[:, :, 1] - F[:, :, 0]                                                         # kernels.py:300-304
            else:
                F = self._nys(Zs.reshape(lt * T, d)).reshape(lt, T, -1)
            Phi, k = [torch.ones((T, 1), dtype=F.dtype, device=F.device)], 0
            for i in range(1, self.mod.kern.num_levels + 1):
                R = F[k]; k += 1
                for j in range(1, i):
                    R = _apply_sketch(self.sk[j - 1], F[k], R); k += 1                              # signature_algs.py:217 / :219
                Phi.append(R)
            self._tens[key] = (Zs, Phi)
        return self._tens[key][1]


def _low_rank_scoped(fn):
    """The low-rank scope of a public evaluation ends with it."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        try:
            return fn(self, *a, **kw)
        finally:
            self._lr = None
            self._phi_memo = None
    return wrapped


class SignatureKernelModule(torch.nn.Module):
    """Trainable view of a ``gpsig_amd.kernels.SignatureKernel`` (exact mode: any order -- the higher-order algorithms'
    gradients run through scratch-based kernels built for coverage rather than speed, grad_ho_kernels.hpp; low-rank mode: torch
    ops, see the module docstring; its evaluations take ``lr=`` -- a ``LowRankDraw`` -- and draw one per evaluation without).

    ``kern`` supplies the structure (base kernel, levels, normalisation, lags, ...) and the initial hyper-parameter values;
    ``write_back()`` copies the trained values into it so that the fused inference path (``kern.K`` etc.) uses them."""

    def __init__(self, kern: SignatureKernel, device="cuda"):
        super().__init__()
        if kern._base is None:
            raise NotImplementedError("SignatureKernel is abstract: use SignatureLinear, SignatureRBF, ...")
        self.kern = kern
        self._lr = None
        self._phi_memo = None
        # linear / cosine kernel: Kzx and the level diagonals from explicit level features (one feature sweep per sequence + plain products).  True:
        # where the recursion kernels' work (tensors x sequences x steps x components x columns) exceeds feature_route_min_work -- a minibatch of 50
        # against 200 tensors is a dozen small launches slower this way (1.2 -> 1.9 ms), BASELINE configs[2] 2.6 times faster; "always"; False
        self.feature_route = True
        self.feature_route_min_work = 3.0e9          # (linear: covariances forward + backward at T = 512, L = 50, d = 6 cross over near 2,000 sequences; round 6)
        # SignatureCosine's recursion kernels are the run-time family (no compile-time instance of the reverse tile kernel): the feature route wins at every size
        # measured (minibatch of 50: 2.3 against 2.7 ms; 4,096 sequences: 2.1 against 51.6)
        self.feature_route_min_work_cosine = 0.0
        self.sum_route = True          # K(X [, X2]) of the linear / cosine kernel: level sum and gradient as one op where the library offers it
        d_cols = kern.num_features * (kern.num_lags + 1)
        # beyond 64 columns and for the spectral kernel: base-kernel tensors here (GEMMs, autograd), recursions in the library
        # (round 6: beyond 64 columns the library's wide route takes the primitives it is built for -- _mx below)
        self.matrix_route = kern._base == "spectral" and not kern.low_rank
        self._d_cols = d_cols
        dev = torch.device(device)
        par = lambda v: torch.nn.Parameter(torch.as_tensor(np.asarray(v, dtype=np.float64), device=dev))
        self.raw_variances = par(positive_inverse(kern.variances))
        self.raw_sigma = par(positive_inverse(kern.sigma))
        self.raw_lengthscales = par(positive_inverse(kern.lengthscales)) if kern.lengthscales is not None else None
        if kern.num_lags > 0:
            self.raw_lags = par(logistic_inverse(kern.lags))
            self.raw_gamma = par(positive_inverse(kern.gamma))
        if kern._base == "spectral":
            self.raw_alpha = par(positive_inverse(kern.alpha))                                      # kernels.py:912-914: transforms.positive
            self.raw_omega = par(positive_inverse(kern.omega))
            self.raw_sgamma = par(positive_inverse(kern.gamma))
        bp = kern._current_base_params()
        self._has_p0 = kern._base in ("poly", "mix")
        self.raw_p0 = par(positive_inverse(bp[0])) if self._has_p0 else None
        self._spec = _Spec(kern._base, kern.num_levels, kern.difference, p1=float(bp[1]) if len(bp) > 1 else 0.0, order=kern.order)
        # load_state_dict() copies through ``.data``-like paths that bump no version counter the weights' host memo is keyed on
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_host_copies())

    # constrained values
    @property
    def variances(self): return positive(self.raw_variances)
    @property
    def sigma(self): return positive(self.raw_sigma)
    @property
    def lengthscales(self): return None if self.raw_lengthscales is None else positive(self.raw_lengthscales)
    @property
    def lags(self): return torch.sigmoid(self.raw_lags)
    @property
    def gamma(self): return positive(self.raw_gamma)
    @property
    def p0(self): return positive(self.raw_p0) if self._has_p0 else None

    def write_back(self):
        k = self.kern
        k.variances = self.variances.detach().cpu().numpy()
        k.sigma = float(self.sigma.detach().cpu())
        if self.raw_lengthscales is not None:
            k.lengthscales = self.lengthscales.detach().cpu().numpy()
        if k.num_lags > 0:
            k.lags, k.gamma = self.lags.detach().cpu().numpy(), self.gamma.detach().cpu().numpy()
        if self._has_p0:
            k._set_base_p0(float(self.p0.detach().cpu()))
        if k._base == "spectral":
            k.alpha, k.omega, k.gamma = (positive(r).detach().cpu().numpy() for r in (self.raw_alpha, self.raw_omega, self.raw_sgamma))
        return k

    # ---- scaling -------------------------------------------------------------------------------------------------
    def _seq3(self, X, presliced=False):
        if not presliced:
            X, _ = self.kern._slice(X, None)                                                        # GPflow Kernel._slice (kernels.py:411-415)
        return X.reshape(X.shape[0], -1, self.kern.num_features)                                    # kernels.py:417-418

    def scale_sequences(self, X):
        """kernels.py:343-364.  (N, L, d) -> (N, L, d * (num_lags + 1))."""
        k = self.kern
        N, L, _ = X.shape
        if k.num_lags > 0:
            X = _add_lags(X, self.lags)                                                             # :353
        X = X.reshape(N, L, k.num_lags + 1, k.num_features)                                         # :355
        if self.raw_lengthscales is not None:
            X = X / self.lengthscales[None, None, None, :]                                          # :358
        if k.num_lags > 0:
            X = X * self.gamma[None, None, :, None]                                                 # :361
        return X.reshape(N, L, -1)

    def scale_tensors(self, Z):
        """kernels.py:367-398 (both layouts: (lt, T, d') and (lt, T, 2, d'))."""
        k = self.kern
        if self.raw_lengthscales is None:
            return Z                                                                                # :374 / :391
        shape = Z.shape
        Z = Z.reshape(*shape[:-1], k.num_lags + 1, k.num_features) / self.lengthscales
        if k.num_lags > 0:
            Z = Z * self.gamma[:, None]
        return Z.reshape(shape)

    # ---- level primitives ------------------------------------------------------------------------------------------
    def _mx(self, prim, cols=0):
        """Does level primitive ``prim`` ('seq', 'diag', 'tens', 'tvs') take the matrix route?  Always when ``matrix_route`` is set (the spectral
        kernel; A/B runs).  Beyond 64 columns the library's exact-shape kernels are not built: the wide route takes what it is built for (the
        distance kernels; sequence lattices of up to 512 columns -- ``cols``: the shorter side's length; order > 1 within the limits below), the matrix route the rest."""
        if self.matrix_route:
            return True
        if self._d_cols <= 64 or self.kern.low_rank:
            return False
        first = self._spec.order == 1 or self._spec.num_levels == 1 or prim == "tens"            # (Kzz has no order)
        # order > 1 on the wide route: the Kzx chains at orders <= 4; the sequence lattices' sweeps (csrc/grad_wave_ho_kernel.hpp) at <= 5 levels, orders <= 4
        high = min(self._spec.order, self._spec.num_levels) <= 4 and (prim == "tvs" or self._spec.num_levels <= 5)
        wide = (prim in WIDE_PRIMITIVES and self._spec.base in WIDE_BASES and (first or high)
                and _WIDE["value"] != 0 and cols - int(self._spec.difference) <= WIDE_LAT_MAX_COLS)
        return not wide

    def _seq_levels(self, Xs, X2s=None):
        if self._lr is not None:                                                                    # kernels.py:426 / :451
            P1 = self._lr.seq(Xs)
            P2 = P1 if X2s is None else self._lr.seq(X2s)
            return torch.stack([a @ b.T for a, b in zip(P1, P2)], dim=0)
        return self._mx_seq_levels(Xs, X2s) if self._mx("seq", (Xs if X2s is None else X2s).shape[1]) else _SeqGramLevels.apply(Xs, X2s, self.p0, self._spec)

    def _phi(self, Xs, work=None):
        """The level features (N, ld) of the scaled sequences where the feature route applies (one sweep per evaluation, shared by the
        level diagonals and Kzx), else None.  work: the caller's estimate of what the recursion kernels would do instead; without one only
        features that the evaluation has already built are handed out."""
        if not (self.feature_route and self._lr is None and not self.matrix_route and self._d_cols <= 64 and self._spec.base in ("linear", "cosine") and Xs.is_cuda
                and Xs.dtype == torch.float64):        # (float32 modules: the recursions' ops convert on the way in and round on the way out)
            return None
        for held, Phi in (self._phi_memo or ()):
            if held is Xs:
                return Phi
        min_work = self.feature_route_min_work_cosine if self._spec.base == "cosine" else self.feature_route_min_work
        if self.feature_route != "always" and self._spec.order == 1 and (work is None or work < min_work):
            return None                                # (order > 1: the higher-order recursion kernels lose at every size -- 3.5 -> 2.0 ms at a minibatch of 50)
        n, l, d = Xs.shape
        ld = _SigFeatures.ld(self._spec, d, l)
        if not ld:
            return None
        # the (n, ld) feature buffer (37k doubles per sequence at d = 8, num_levels = 5) must fit beside what is already allocated: a large-n Kdiag
        # or normalisation that needs O(n * levels) memory through the recursion kernels must not run out through this route (whatever the order)
        try:
            free_b, _ = torch.cuda.mem_get_info(Xs.device)
            # + what torch's caching allocator holds but has not handed out: a warmed-up training process has reserved most of HBM, the driver's
            # figure alone would send every linear / cosine evaluation to the pair recursion for good
            free_b += max(0, torch.cuda.memory_reserved(Xs.device) - torch.cuda.memory_allocated(Xs.device))
        except Exception:  # noqa: BLE001
            free_b = None
        if free_b is not None and 2.5 * 8.0 * n * ld > free_b:     # features, their gradient and a working copy
            return None
        Phi = _SigFeatures.apply(Xs, self._spec)
        self._phi_memo = ((self._phi_memo or ())[-1:]) + ((Xs, Phi),)         # the evaluation's last two sets of sequences
        return Phi

    def _tvs_work(self, Zs, Xs):
        """tensors x sequences x time steps x tensor components x columns: what the tensor-vs-sequence recursion kernels sweep."""
        return float(Zs.shape[1]) * Xs.shape[0] * Xs.shape[1] * Zs.shape[0] * Xs.shape[2]

    def _diag_levels(self, Xs):
        if self._lr is not None:                                                                    # kernels.py:457, :501
            return torch.stack([torch.square(P).sum(dim=-1) for P in self._lr.seq(Xs)], dim=0)
        Phi = self._phi(Xs)
        if Phi is not None:                                                                         # K_m(x, x) = |Phi_m(x)|^2
            return _LevelNorms.apply(Phi, Xs.shape[2], self._spec.num_levels)
        return self._mx_diag_levels(Xs) if self._mx("diag", Xs.shape[1]) else _SeqDiagLevels.apply(Xs, self.p0, self._spec)

    def _tens_levels(self, Zs, increments):
        if self._lr is not None:                                                                    # kernels.py:525-527
            return torch.stack([P @ P.T for P in self._lr.tens(Zs, increments)], dim=0)
        return self._mx_tens_levels(Zs, increments) if self._mx("tens") else _TensGramLevels.apply(Zs, self.p0, self._spec, increments)

    def _tvs_levels(self, Zs, Xs, increments):
        if self._lr is not None:                                                                    # kernels.py:568
            return torch.stack([a @ b.T for a, b in zip(self._lr.tens(Zs, increments), self._lr.seq(Xs))], dim=0)
        Phi = self._phi(Xs, self._tvs_work(Zs, Xs))
        if Phi is not None:                                                                         # K_m(z, x) = <z_1 (x) .. (x) z_m, Phi_m(x)>
            lev = _split_levels(Phi, Xs.shape[2], self._spec.num_levels)
            zf = _tensor_features(Zs, self._spec.num_levels, increments, self._spec.base == "cosine")
            ones = torch.ones((zf[0].shape[0], Xs.shape[0]), dtype=lev[0].dtype, device=Xs.device)
            return torch.stack([ones] + [a @ b.T for a, b in zip(zf, lev)], dim=0)
        return self._mx_tvs_levels(Zs, Xs, increments) if self._mx("tvs") else _TensVsSeqLevels.apply(Zs, Xs, self.p0, self._spec, increments)

    def _tvs_weighted(self, Zs, Xs, fac, increments):
        if self._mx("tvs") or self._lr is not None:
            return (self._tvs_levels(Zs, Xs, increments) * fac[:, None, :]).sum(dim=0)
        Phi = self._phi(Xs, self._tvs_work(Zs, Xs))
        if Phi is not None:             # sum_m fac[m][n] <Z_m[t], Phi_m[n]>: the factors go onto the features, no level arrays
            # ONE product of depth ld (the library's per-level GEMMs of depth d, d^2 run at the speed of the widest): level 0 (= 1 on both sides)
            # rides along as the column it has in the feature buffer, whose zero padding keeps the rows 16-aligned
            zl = _tensor_features(Zs, self._spec.num_levels, increments, self._spec.base == "cosine")
            F, T = sum(a.shape[1] for a in zl), zl[0].shape[0]
            zf = torch.cat(zl + [zl[0].new_ones(T, 1), zl[0].new_zeros(T, Phi.shape[1] - F - 1)], dim=1)
            return _FeatureProduct.apply(zf, _ScaleLevels.apply(Phi, fac, Xs.shape[2]))
        return _TensVsSeqWeighted.apply(Zs, Xs, fac, self.p0, self._spec, increments)

    # the same four primitives on the matrix route (kernels.py:188-340 with torch ops up to the differenced tensor)
    def _kappa(self, A, B):
        spec = None
        if self.kern._base == "spectral":
            spec = (self.kern.family, positive(self.raw_alpha), positive(self.raw_omega), positive(self.raw_sgamma))
        return base_kernel_matrix(self.kern._base, A, B, self.p0, self._spec.p1, spec)

    def _mx_lattices(self, K4):
        """(P, L1, L2) base-kernel lattices -> levels (M+1, P): signature_algs.py:25-26 here, :28-74 in the library."""
        if self.kern.difference:
            K4 = K4[:, 1:, 1:] + K4[:, :-1, :-1] - K4[:, :-1, 1:] - K4[:, 1:, :-1]
        return _LatticeLevels.apply(K4.contiguous(), self._spec)

    def _mx_seq_levels(self, Xs, X2s=None):
        N1, L1, d = Xs.shape                                                                        # kernels.py:208-237
        Ys = Xs if X2s is None else X2s
        N2, L2 = Ys.shape[:2]
        Kt = self._kappa(Xs.reshape(N1 * L1, d), Ys.reshape(N2 * L2, d)).reshape(N1, L1, N2, L2)
        return self._mx_lattices(Kt.permute(0, 2, 1, 3).reshape(N1 * N2, L1, L2)).reshape(-1, N1, N2)

    def _mx_diag_levels(self, Xs):
        return self._mx_lattices(self._kappa(Xs, Xs))                                               # kernels.py:188-205

    def _mx_tens_levels(self, Zs, increments):
        lt, T, d = Zs.shape[0], Zs.shape[1], Zs.shape[-1]                                           # kernels.py:263-283
        if increments:
            Mk = self._kappa(Zs.reshape(lt, 2 * T, d), Zs.reshape(lt, 2 * T, d)).reshape(lt, T, 2, T, 2)
            Mk = Mk[:, :, 1, :, 1] + Mk[:, :, 0, :, 0] - Mk[:, :, 1, :, 0] - Mk[:, :, 0, :, 1]      # :276-277
        else:
            Mk = self._kappa(Zs, Zs)
        lev, k = [torch.ones_like(Mk[0])], 0                                                        # signature_algs.py:76-99
        for i in range(1, self._spec.num_levels + 1):
            R = Mk[k]
            for j in range(1, i):
                R = R * Mk[k + j]
            lev.append(R)
            k += i
        return torch.stack(lev, dim=0)

    def _mx_tvs_levels(self, Zs, Xs, increments):
        lt, T, d = Zs.shape[0], Zs.shape[1], Zs.shape[-1]                                           # kernels.py:313-340
        N, L = Xs.shape[:2]
        Xf = Xs.reshape(N * L, d)
        if increments:
            Mk = self._kappa(Zs.reshape(lt * T * 2, d), Xf).reshape(lt, T, 2, N, L)
            Mk = Mk[:, :, 1] - Mk[:, :, 0]                                                          # :329-330
        else:
            Mk = self._kappa(Zs.reshape(lt * T, d), Xf).reshape(lt, T, N, L)
        if self.kern.difference:
            Mk = Mk[..., 1:] - Mk[..., :-1]                                                         # signature_algs.py:114
        if self._spec.order > 1 and self._spec.num_levels > 1:
            return self._mx_tvs_chains_higher_order(Mk)
        m = Mk.permute(0, 3, 1, 2).reshape(lt, Mk.shape[-1], T * N)                                 # (lt, R, P), pair index fastest
        return _ChainLevels.apply(m.contiguous(), self._spec).reshape(-1, T, N)

    def _mx_tvs_chains_higher_order(self, Mk):
        """signature_algs.py:144-158 on the differenced tensor Mk (lt, T, N, R), array operation by array operation with torch's autograd behind it: the
        corner the library's kernels do not take -- order > 1 on the matrix route, i.e. more than 64 columns (or the spectral kernel) with a base kernel
        outside the wide route's families (those run wide_tvs_fwd / _bwd_kernel at any width).  Built for coverage."""
        def excumsum(A):                                                                            # tf.cumsum(exclusive=True, axis=2)
            return torch.cumsum(A, dim=2) - A
        order, lev, k = self._spec.order, [torch.ones_like(Mk[0, :, :, 0])], 0                      # :144
        for i in range(1, self._spec.num_levels + 1):                                               # :147
            R = [Mk[k]]                                                                             # :148
            k += 1
            for j in range(1, i):                                                                   # :150
                dcur = min(j + 1, order)                                                            # :151
                Rn = [Mk[k] * excumsum(sum(R))]                                                     # :153
                for l in range(1, dcur):                                                            # :154
                    Rn.append(Mk[k] * R[l - 1] / float(l + 1))                                      # :155
                R = Rn
                k += 1
            lev.append(sum(R).sum(dim=2))                                                           # :158
        return torch.stack(lev, dim=0)

    def _w(self):
        return self.sigma * self.variances                                                          # kernels.py:471

    def _w_host(self, w):
        """The level weights on the host for the one-op level sum (the C ABI takes them by value): the copy is a blocking device-to-host
        synchronisation, so it is made once per VALUE of the parameters -- keyed on their tensors' version counters, which every in-place
        optimiser update bumps -- instead of once per forward pass."""
        # Writes through ``.data`` (p.data.fill_(), p.data.copy_()) bump no version counter: load_state_dict() drops the memo through the hook
        # registered in __init__; after any other ``.data`` write call ``invalidate_host_copies()``.
        key = (self.raw_sigma.data_ptr(), self.raw_sigma._version, self.raw_variances.data_ptr(), self.raw_variances._version)
        held = getattr(self, "_w_host_memo", None)
        if held is None or held[0] != key:
            held = (key, _SeqGramSum.weights_on_host(w))
            self._w_host_memo = held
        return held[1]

    def invalidate_host_copies(self):
        """Forget the host copies of parameter values (the level weights the C ABI takes by value).  Needed only after writing parameters through
        ``.data`` -- in-place updates through autograd-visible ops (every optimiser step) and load_state_dict() are noticed."""
        self._w_host_memo = None

    # ---- low-rank mode ---------------------------------------------------------------------------------------------
    def draw_low_rank(self, num_points):
        """The value-independent random objects of one evaluation over ``num_points`` points (``kern.rng``, as ``kern.draw_low_rank``):
        landmark indices, the jitter draw (low_rank_calculations.py:52), one projection per level >= 2."""
        from . import low_rank as _lrm
        k = self.kern
        c = int(k.num_components)
        if c > num_points:
            raise ValueError("num_components exceeds the number of available points")
        idx = np.sort(k.rng.choice(int(num_points), size=c, replace=False, shuffle=False))
        return LowRankDraw(idx, JITTER * k.rng.random(c), _lrm.draw_level_sketches(k.rng, k.num_levels, c, int(k.rank_bound), k.sparsity))

    def _lr_open(self, lr, *points):
        """Low-rank mode: gather the landmarks from the concatenation of the evaluation's scaled points (the order the reference
        concatenates in: tensors first, kernels.py:562-563, :614-615; X before X2, :445-446) and whiten them; a fresh draw per
        evaluation unless one is handed in."""
        if not self.kern.low_rank:
            if lr is not None:
                raise ValueError("lr= is for kernels in low-rank mode")
            return
        if self._spec.order > 1 and self.kern.num_levels > 1:
            raise NotImplementedError('Higher-order algorithms not compatible with low-rank mode (yet).')   # kernels.py:59-60
        if not all(p_.is_cuda for p_ in points):
            raise RuntimeError("gpsig_amd.autodiff needs CUDA (ROCm) tensors: there is no CPU path")
        pool = torch.cat([p_.reshape(-1, p_.shape[-1]) for p_ in points], dim=0)
        self._lr = _LowRankScope(self, pool, lr if lr is not None else self.draw_low_rank(pool.shape[0]))

    # ---- kernel evaluations ----------------------------------------------------------------------------------------
    @_low_rank_scoped
    def K(self, X, X2=None, presliced=False, return_levels=False, presliced_X=False, presliced_X2=False, lr=None):
        """kernels.py:401-476.  lr: a LowRankDraw (low-rank mode; drawn per evaluation when None)."""
        Xs = self.scale_sequences(self._seq3(X, presliced or presliced_X))
        N = Xs.shape[0]
        X2s = None if X2 is None else self.scale_sequences(self._seq3(X2, presliced or presliced_X2))
        if (self.sum_route and not return_levels and not self.kern.low_rank and not self.matrix_route and self._d_cols <= 64 and lr is None
                and self._spec.base in ("linear", "cosine") and Xs.is_cuda and not torch.cuda.is_current_stream_capturing()):
            # the linear / cosine kernel's level sum and its gradient as one op through the feature space (no level arrays)
            if _SeqGramSum.applies(Xs, X2s, self._spec, self.kern.normalization):
                w = self._w()
                return _SeqGramSum.apply(Xs, X2s, w, self._w_host(w), self._spec, self.kern.normalization)
        if X2 is None:
            self._lr_open(lr, Xs)
            K = self._seq_levels(Xs)
            if self.kern.normalization:
                K = K + JITTER * torch.eye(N, dtype=K.dtype, device=K.device)[None]                 # :431
                dsq = torch.sqrt(torch.diagonal(K, dim1=1, dim2=2))                                 # :432
                K = K / (dsq[:, :, None] * dsq[:, None, :])                                         # :433
        else:
            self._lr_open(lr, Xs, X2s)
            K = self._seq_levels(Xs, X2s)
            if self.kern.normalization:
                d1 = torch.sqrt(self._diag_levels(Xs) + JITTER)                                     # :460-466
                d2 = torch.sqrt(self._diag_levels(X2s) + JITTER)
                K = K / (d1[:, :, None] * d2[:, None, :])                                           # :469
        K = K * self._w()[:, None, None]
        return K if return_levels else K.sum(dim=0)

    @_low_rank_scoped
    def Kdiag(self, X, presliced=False, return_levels=False, lr=None):
        """kernels.py:479-510."""
        N = X.shape[0]
        if self.kern.normalization:
            Kd = self._w()[:, None].expand(-1, N)                                                   # :486-490
        else:
            Xs = self.scale_sequences(self._seq3(X, presliced))
            self._lr_open(lr, Xs)
            Kd = self._diag_levels(Xs) * self._w()[:, None]
        return Kd if return_levels else Kd.sum(dim=0)

    @_low_rank_scoped
    def K_tens(self, Z, return_levels=False, increments=False, lr=None):
        """kernels.py:513-536."""
        Zs = self.scale_tensors(Z)
        self._lr_open(lr, Zs)
        K = self._tens_levels(Zs, increments) * self._w()[:, None, None]
        return K if return_levels else K.sum(dim=0)

    @_low_rank_scoped
    def K_tens_vs_seq(self, Z, X, return_levels=False, increments=False, presliced=False, lr=None):
        """kernels.py:539-588."""
        Xs = self.scale_sequences(self._seq3(X, presliced))
        Zs0 = self.scale_tensors(Z)
        self._lr_open(lr, Zs0, Xs)
        self._phi(Xs, self._tvs_work(Zs0, Xs))                                                      # (so that the level diagonals share the feature sweep)
        if not return_levels:
            # the same numbers with the level sum taken inside the kernel: (M+1, N) factors in, (T, N) out
            fac = self._w()[:, None].expand(-1, Xs.shape[0])                                        # :584
            if self.kern.normalization:
                fac = fac / torch.sqrt(self._diag_levels(Xs) + JITTER)                              # :576-581
            return self._tvs_weighted(Zs0, Xs, fac, increments)                                     # :588
        K = self._tvs_levels(Zs0, Xs, increments)
        if self.kern.normalization:
            K = K / torch.sqrt(self._diag_levels(Xs) + JITTER)[:, None, :]                          # :576-581
        return K * self._w()[:, None, None]

    @_low_rank_scoped
    def K_tens_n_seq_covs(self, Z, X, full_X_cov=False, return_levels=False, increments=False, presliced=False, lr=None):
        """kernels.py:591-671: Kzz, Kzx and Kxx (full or diagonal) from one scaling of the inputs."""
        Xs = self.scale_sequences(self._seq3(X, presliced))
        N = Xs.shape[0]
        Zs = self.scale_tensors(Z)
        self._lr_open(lr, Zs, Xs)
        self._phi(Xs, self._tvs_work(Zs, Xs))                                                       # (so that the level diagonals share the feature sweep)
        Kzz = self._tens_levels(Zs, increments)                                                     # :623
        w = self._w()
        if not return_levels:
            # Kzx as a weighted level sum taken inside the kernel (no (M+1, T, N) array in a training step): the factors are what
            # :638 / :660 divide by and :667 multiplies with
            if full_X_cov:
                Kxx = self._seq_levels(Xs)                                                          # :630
                fac = w[:, None].expand(-1, N)
                if self.kern.normalization:
                    Kxx = Kxx + JITTER * torch.eye(N, dtype=Kxx.dtype, device=Kxx.device)[None]     # :633
                    dsq = torch.sqrt(torch.diagonal(Kxx, dim1=1, dim2=2))
                    Kxx = Kxx / (dsq[:, :, None] * dsq[:, None, :])                                 # :637
                    fac = fac / dsq                                                                 # :638
                Kxx = (Kxx * w[:, None, None]).sum(dim=0)
            else:
                dl = self._diag_levels(Xs)                                                          # :653
                if self.kern.normalization:
                    fac = w[:, None] / torch.sqrt(dl + JITTER)                                      # :656-660
                    Kxx = w.sum().expand(N)                                                         # :661
                else:
                    fac = w[:, None].expand(-1, N)
                    Kxx = (dl * w[:, None]).sum(dim=0)
            return (Kzz * w[:, None, None]).sum(dim=0), self._tvs_weighted(Zs, Xs, fac, increments), Kxx
        Kzx = self._tvs_levels(Zs, Xs, increments)                                                  # :624
        if full_X_cov:
            Kxx = self._seq_levels(Xs)                                                              # :630
            if self.kern.normalization:
                Kxx = Kxx + JITTER * torch.eye(N, dtype=Kxx.dtype, device=Kxx.device)[None]         # :633
                dsq = torch.sqrt(torch.diagonal(Kxx, dim1=1, dim2=2))
                Kxx = Kxx / (dsq[:, :, None] * dsq[:, None, :])                                     # :637
                Kzx = Kzx / dsq[:, None, :]                                                         # :638
            Kxx = Kxx * w[:, None, None]
        else:
            Kxx = self._diag_levels(Xs)                                                             # :653
            if self.kern.normalization:
                Kzx = Kzx / torch.sqrt(Kxx + JITTER)[:, None, :]                                    # :656-660
                Kxx = w[:, None].expand(-1, N)                                                      # :661
            else:
                Kxx = Kxx * w[:, None]
        return Kzz * w[:, None, None], Kzx * w[:, None, None], Kxx

    @_low_rank_scoped
    def K_seq_n_seq_covs(self, X, X2, full_X2_cov=False, return_levels=False, presliced=False, lr=None):
        """kernels.py:674-761 (``X`` = inducing sequences, never sliced: :679-680; ``X2`` = data), including the double division
        of :713 + :750."""
        Xs = self.scale_sequences(self._seq3(X, True))
        X2s = self.scale_sequences(self._seq3(X2, presliced))
        self._lr_open(lr, Xs, X2s)
        N, N2 = Xs.shape[0], X2s.shape[0]
        w = self._w()
        Kxx = self._seq_levels(Xs)
        norm = self.kern.normalization
        # linear / cosine kernel, level sum wanted: Kxx2 = sum_m facz[m][t] facx[m][n] <Phi_m(z_t), Phi_m(x_n)> as ONE product of scaled level features
        Pz = Px = None
        if not return_levels:
            work = float(N) * N2 * Xs.shape[1] * X2s.shape[1] * Xs.shape[2]                        # pairs x lattice cells x columns
            Pz = self._phi(Xs, work)
            Px = self._phi(X2s, work) if Pz is not None else None
        by_features = Px is not None
        Kxx2 = None if by_features else self._seq_levels(Xs, X2s)
        facz, facx = w[:, None].expand(-1, N), None
        if norm:
            Kxx = Kxx + JITTER * torch.eye(N, dtype=Kxx.dtype, device=Kxx.device)[None]             # :709
            dsq = torch.sqrt(torch.diagonal(Kxx, dim1=1, dim2=2))
            Kxx = Kxx / (dsq[:, :, None] * dsq[:, None, :])
            if by_features:
                facz = facz / dsq
            else:
                Kxx2 = Kxx2 / dsq[:, :, None]                                                       # :713
        if full_X2_cov:
            Kx2x2 = self._seq_levels(X2s)
            if norm:
                Kx2x2 = Kx2x2 + JITTER * torch.eye(N2, dtype=Kxx.dtype, device=Kxx.device)[None]
                d2 = torch.sqrt(torch.diagonal(Kx2x2, dim1=1, dim2=2))
                if by_features:
                    facx = 1.0 / d2
                else:
                    Kxx2 = Kxx2 / d2[:, None, :]
                Kx2x2 = Kx2x2 / (d2[:, :, None] * d2[:, None, :])
            Kx2x2 = Kx2x2 * w[:, None, None]
        else:
            Kx2x2 = self._diag_levels(X2s)
            if norm:
                d2 = torch.sqrt(Kx2x2 + JITTER)
                if by_features:
                    facz, facx = facz / dsq, 1.0 / d2                                               # :750 (second division by dsq: reference quirk)
                else:
                    Kxx2 = Kxx2 / (dsq[:, :, None] * d2[:, None, :])                                # :750 (second division by dsq: reference quirk)
                Kx2x2 = w[:, None].expand(-1, N2)
            else:
                Kx2x2 = Kx2x2 * w[:, None]
        Kxx = Kxx * w[:, None, None]
        if by_features:
            d = Xs.shape[2]
            A = _ScaleLevels.apply(Pz, facz, d)
            B = Px if facx is None else _ScaleLevels.apply(Px, facx, d)
            return Kxx.sum(dim=0), _FeatureProduct.apply(A, B), Kx2x2.sum(dim=0)
        Kxx2 = Kxx2 * w[:, None, None]
        if return_levels:
            return Kxx, Kxx2, Kx2x2
        return Kxx.sum(dim=0), Kxx2.sum(dim=0), Kx2x2.sum(dim=0)
