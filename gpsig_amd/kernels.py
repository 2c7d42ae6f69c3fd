"""Signature kernels on MI355X: the call surface of ``gpsig.kernels`` (reference: gpsig/kernels.py).

Same class names, constructor arguments, validation errors and array conventions as the reference's
``SignatureKernel`` family; the work is done by the HIP library behind ``include/gpsig_hip.h``.
TensorFlow / GPflow are not involved: hyper-parameters are plain NumPy attributes holding the
constrained values (``variances`` (M+1,), ``sigma`` scalar, ``lengthscales`` (d,) or None, ...), and
every method is numpy-in / numpy-out (host pointers) or torch-CUDA-in / torch-CUDA-out (device
pointers, asynchronous on the current stream).

float32 inputs are computed in float32 where the float32 kernels are built, else by the float64 kernels and rounded.
Not built (raise NotImplementedError, never a silent CPU fallback): ``SignatureSpectral`` in low-rank mode.  Training (gradients): ``gpsig_amd.autodiff``.
"""
import ctypes as C

import sys as _sys
import warnings as _warnings

import numpy as np

from . import _lib, low_rank as _lr

try:  # torch is only plumbing: device memory and streams
    import torch
except Exception:  # pragma: no cover
    torch = None

JITTER = 1e-6   # gpflow.settings.jitter
_PINNED_OUT_MIN = 1 << 20   # host-array results of at least this many bytes are allocated page-locked


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


class _Launch:
    """Marshals one C-ABI call: pointer mode, contiguity, output allocation."""

    def __init__(self, *arrays):
        tens = [a for a in arrays if a is not None and _is_torch(a) and a.is_cuda]
        self.device_mode = len(tens) > 0
        if self.device_mode:
            if any(a is not None and not (_is_torch(a) and a.is_cuda) for a in arrays):
                raise ValueError("mixing CUDA tensors with host arrays in one call")
            self.dev = tens[0].device
            stream = torch.cuda.current_stream(self.dev).cuda_stream
            self.ctx = _lib.context(self.dev.index or 0, stream)
            self.ctx.set_pointer_mode(_lib.PTR_DEVICE)
        else:
            self.ctx = _lib.context(0, 0)
            self.ctx.set_pointer_mode(_lib.PTR_HOST)
        self.keep = []
        # the arithmetic type follows the data (the reference has one global settings.float_type): float32 in,
        # float32 kernels and float32 out; anything else is computed in float64
        given = [a for a in arrays if a is not None]
        self.f32 = len(given) > 0 and all(str(a.dtype).endswith("float32") for a in given)
        self.np_dtype = np.float32 if self.f32 else np.float64
        self.dtype_id = _lib.F32 if self.f32 else _lib.F64

    def inp(self, a):
        if a is None:
            return None
        if self.device_mode:
            t = a.detach().to(torch.float32 if self.f32 else torch.float64).contiguous()
            self.keep.append(t)
            return C.c_void_p(t.data_ptr())
        t = np.ascontiguousarray(a.detach().cpu().numpy() if _is_torch(a) else a, dtype=self.np_dtype)
        self.keep.append(t)
        return C.c_void_p(t.ctypes.data)

    def out(self, shape):
        if self.device_mode:
            t = torch.empty(tuple(int(s) for s in shape), dtype=torch.float32 if self.f32 else torch.float64, device=self.dev)
            return t, C.c_void_p(t.data_ptr())
        shape = tuple(int(s) for s in shape)
        if torch is not None and int(np.prod(shape)) * np.dtype(self.np_dtype).itemsize >= _PINNED_OUT_MIN:
            # A large result in page-locked memory from torch's caching host allocator: the DMA engines write it directly (134 MB in
            # 2.4 ms), where a fresh np.empty costs 10 ms of first-touch page faults on top of the copy (tools/bench_host_e2e.py).
            # The array is an ordinary ndarray to the caller; its memory goes back to the allocator's pool with its last reference.
            try:
                t = torch.empty(shape, dtype=torch.float32 if self.f32 else torch.float64, pin_memory=True).numpy()
                return t, C.c_void_p(t.ctypes.data)
            except Exception:       # no page-locked memory to be had: pageable memory through the library's bounce buffers
                pass
        t = np.empty(shape, dtype=self.np_dtype)
        return t, C.c_void_p(t.ctypes.data)


class GraphedCall:
    """A recorded evaluation: `inputs` are the tensors it reads (refresh them in place), `out` what it writes."""

    def __init__(self, graph, inputs, out, stream):
        self.graph, self.inputs, self.out, self.stream = graph, inputs, out, stream

    def close(self):
        """Release the recording: waits for replays in flight, destroys the executable graph, gives the side stream's context back.
        Errors surface here; __del__ calls it and reports instead of raising (a destructor cannot)."""
        if self.graph is None:
            return
        dev = self.out[0].device if isinstance(self.out, (tuple, list)) else self.out.device
        self.stream.synchronize()            # no replay may still be in flight when the executable graph goes away
        g, self.graph = self.graph, None
        g.destroy()
        _lib.release(dev.index or 0, self.stream.cuda_stream)      # the side stream's context belongs to this recording alone

    def __del__(self):
        if _sys is None or _sys.meta_path is None or _sys.is_finalizing():
            return                           # interpreter shutdown: the process gives the device back; modules this needs are already gone
        try:
            self.close()
        except Exception as e:               # not silently: a failed release leaks a context and its scratch buffers
            try:
                _warnings.warn("GraphedCall: releasing a recorded evaluation failed: %r" % (e,), ResourceWarning)
            except Exception:
                pass

    def replay(self):
        cur = torch.cuda.current_stream(self.out[0].device if isinstance(self.out, (tuple, list)) else self.out.device)
        self.stream.wait_stream(cur)             # the inputs were refreshed on the caller's stream
        self.graph.launch()
        cur.wait_stream(self.stream)
        return self.out


def _shape(a):
    return tuple(a.shape)


def _launch_f64(*arrays):
    """_Launch for the low-rank entry points, which are built for float64 only: all-float32 arguments raise (and
    _f32_upcast retries the call in float64) instead of handing float32 buffers to entry points that read doubles."""
    L_ = _Launch(*arrays)
    if L_.f32:
        raise NotImplementedError("low-rank mode is built for float64 only")
    return L_


def _is_f32(a):
    return a is not None and hasattr(a, "dtype") and str(a.dtype).endswith("float32")


def _f32_upcast(method):
    """float32 evaluations the float32 kernels are not built for (shapes beyond the wavefront kernels, the spectral kernel,
    low-rank mode) are computed by the float64 kernels on the GPU and rounded to float32 -- never less accurate than asked for.
    So are float32 evaluations on ONE-COLUMN state spaces (num_features == 1; round 5): there the level values of a sequence are
    sums of products of scalar increments that cancel by orders of magnitude, and the float32 kernels missed the float32 tolerance
    (1.2e-4 .. 5.6e-3 against 1e-4 in round 4's sweeps, profiles/r04_fuzz.txt; fixtures tests/golden/fuzz_cases.npz) -- every miss the
    sweeps found was of this class, and a one-column problem is small.  And float32 evaluations of SignatureCosine (round 5's sweeps: 1.7e-4 ..
    9e-3, profiles/r05_fuzz.txt): the kernel is scale-free, so sequences away from the origin have cosines within 1e-3 of one another and their
    double increments cancel in float32; where the feature route applies such requests were computed in float64 already."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        arrays = [a for a in args if hasattr(a, "dtype") and hasattr(a, "shape")]
        all_f32 = bool(arrays) and all(_is_f32(a) for a in arrays)
        # (round 6) SignatureRBF at order > 1: the float64 evaluation kernel has exact instances (levels and order at compile time: 14 ms for 2,048 sequences at
        # order 2), the float32 one only run-time ones (44 ms) -- such Grams are computed in float64 and rounded, where those instances exist
        ho = False
        if all_f32 and method.__name__ == "K" and getattr(self, "_base", None) == "rbf" and not getattr(self, "low_rank", False):
            M_, o_ = int(getattr(self, "num_levels", 0)), int(getattr(self, "order", 1))
            o_, de = min(o_, M_), int(getattr(self, "num_features", 0)) * (int(getattr(self, "num_lags", 0) or 0) + 1)
            ho = (o_ == 2 and 3 <= M_ <= 5 and de <= 8) or (o_ in (3, 4) and M_ in (4, 5) and 4 < de <= 8)
        if not (all_f32 and (getattr(self, "num_features", 0) == 1 or getattr(self, "_base", None) == "cosine" or ho)):
            try:
                return method(self, *args, **kwargs)
            except NotImplementedError:
                if not all_f32:
                    raise
        if getattr(self, "_graph_recording", False):
            # graphed(): the float64 copies, the float64 result and its rounding would be torch temporaries recorded inside the capture and
            # recycled after it -- a replay would write into whatever lives there then
            raise ValueError("graphed(): this float32 evaluation is computed in float64 and rounded (one-column state space, SignatureCosine or "
                             "a shape the float32 kernels are not built for); record it with float64 tensors")
        up = lambda a: (a.double() if _is_torch(a) else np.asarray(a, dtype=np.float64)) if _is_f32(a) else a   # noqa: E731
        out = method(self, *[up(a) for a in args], **kwargs)
        down = lambda o: o.float() if _is_torch(o) else np.asarray(o, dtype=np.float32)                          # noqa: E731
        return tuple(down(o) for o in out) if isinstance(out, tuple) else down(out)
    return wrapper


class LowRankState:
    """The random objects of one low-rank evaluation (reference: drawn inside the TF graph, kernels.py:443-449,
    low_rank_calculations.py:47-57): landmarks (c, d') -- scaled points --, whitening (c, c), one sketch per level >= 2."""

    def __init__(self, landmarks, whitening, sketches, rank_bound, jitter_diag=None, eigenvalues=None):
        self.landmarks = np.ascontiguousarray(landmarks, dtype=np.float64)
        self.whitening = np.ascontiguousarray(whitening, dtype=np.float64)
        self.jitter_diag = jitter_diag            # the draw of low_rank_calculations.py:52 the whitening was computed with
        self.eigenvalues = eigenvalues            # of the jittered landmark Gram, ascending (before :56 adds the jitter)
        self.sketches = list(sketches)
        self.rank_bound = int(rank_bound)
        self.num_components = self.landmarks.shape[0]

    def as_c(self, keep):
        arr = (_lib.SketchC * max(len(self.sketches), 1))()
        for k, sk in enumerate(self.sketches):
            arr[k].k1, arr[k].k2, arr[k].r, arr[k].nnz = sk.k1, sk.k2, sk.r, int(sk.val.shape[0])
            arr[k].colptr = sk.colptr.ctypes.data_as(C.POINTER(C.c_int32))
            arr[k].i1 = sk.i1.ctypes.data_as(C.POINTER(C.c_int32))
            arr[k].i2 = sk.i2.ctypes.data_as(C.POINTER(C.c_int32))
            arr[k].val = sk.val.ctypes.data_as(C.POINTER(C.c_double))
        lr = _lib.LowRankC()
        lr.num_components, lr.rank_bound, lr.num_sketches = self.num_components, self.rank_bound, len(self.sketches)
        lr.landmarks = self.landmarks.ctypes.data_as(C.POINTER(C.c_double))
        lr.whitening = self.whitening.ctypes.data_as(C.POINTER(C.c_double))
        lr.sketches = arr
        keep.extend([arr, lr, self])
        return lr


class DeviceLowRankState:
    """The random objects of one low-rank evaluation drawn and kept ON THE DEVICE (gpsig_lr_draw): what the reference draws inside its
    TF graph at every evaluation.  ``export()`` copies them out as a host-side LowRankState (the CPU restatement then evaluates the very
    same objects); a state is drawn into again by the next ``draw_low_rank`` of its kernel object."""

    SPARSITY = {'sqrt': 0, 'log': 1, 'lin': 2}

    def __init__(self, ctx, num_components, rank_bound, num_sketches):
        self.ctx, self._h = ctx, C.c_void_p()
        self._lib = ctx._lib                      # (the library handle outlives the context object's own handle)
        self.num_components, self.rank_bound, self.num_sketches = int(num_components), int(rank_bound), int(num_sketches)

    def close(self):
        """Free the device block.  Safe after the context was closed: gpsig_ctx_destroy detaches the states drawn on it, and a
        detached state only frees its memory (it never touches the context's stream again)."""
        if self._h:
            h, self._h = self._h, C.c_void_p()
            self._lib.gpsig_lr_state_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def as_c(self, keep):
        lr = _lib.LowRankC()
        lr.num_components, lr.rank_bound, lr.num_sketches = self.num_components, self.rank_bound, self.num_sketches
        lr.device_state = self._h
        keep.extend([lr, self])
        return lr

    def export(self):
        """Host copies of what was drawn, as a LowRankState (waits for the stream).  Raises if the draw failed (eigensolver not
        converged / a projection over its capacity: the evaluations from such a state are NaN) or the context was closed."""
        lib, ctx = self.ctx._lib, self.ctx
        if getattr(ctx, "_h", None) is None:
            raise RuntimeError("the library context this low-rank state was drawn on has been closed")
        sizes, nnz = (C.c_int32 * 5)(), (C.c_int32 * max(self.num_sketches, 1))()
        ctx.check(lib.gpsig_lr_state_sizes(ctx._h, self._h, sizes, nnz))
        c, d_eff, r, nsk, self.jacobi_sweeps = (int(v) for v in sizes)
        S, jd, Wh, ev = np.empty((c, d_eff)), np.empty(c), np.empty((c, c)), np.empty(c)
        arr = (_lib.SketchC * max(nsk, 1))()
        host = []
        k2 = c
        for i in range(nsk):
            n = int(nnz[i])
            colptr, i1, i2, val = np.empty(r + 1, np.int32), np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n)
            arr[i].k1, arr[i].k2, arr[i].r, arr[i].nnz = c, k2, r, n
            arr[i].colptr = colptr.ctypes.data_as(C.POINTER(C.c_int32))
            arr[i].i1 = i1.ctypes.data_as(C.POINTER(C.c_int32))
            arr[i].i2 = i2.ctypes.data_as(C.POINTER(C.c_int32))
            arr[i].val = val.ctypes.data_as(C.POINTER(C.c_double))
            host.append((c, k2, r, colptr, i1, i2, val))
            k2 = r
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))                    # noqa: E731
        ctx.check(lib.gpsig_lr_state_export(ctx._h, self._h, dp(S), dp(jd), dp(Wh), dp(ev), arr))
        return LowRankState(S, Wh, [_lr.Sketch(*h) for h in host], r, jitter_diag=jd, eigenvalues=ev)


class SignatureKernel:
    """Reference: ``gpsig.kernels.SignatureKernel`` (gpsig/kernels.py:15-761).

    # Args (as the reference, kernels.py:18-51)
    :input_dim:      total size of one input row = len_examples * num_features
    :num_features:   state-space dimension of the sequences
    :num_levels:     truncation level M of the signature
    :active_dims:    columns of the input that feed the kernel (GPflow ``Kernel._slice``); default all
    :variances:      (M+1,) level variances
    :lengthscales:   (num_features,) or None for no scaling
    :order:          1..M; <=0 or >=M means M (kernels.py:57)
    :normalization:  normalise each level (kernels.py:430-433, :455-469)
    :difference:     difference the base-kernel tensor (signature_algs.py:25-26)
    :num_lags:       None or a non-negative int (kernels.py:70-82)
    :low_rank, num_components, rank_bound, sparsity: validated as the reference does (low-rank mode: gpsig_amd.low_rank, float64)
    """
    _base = None

    def __init__(self, input_dim, num_features, num_levels, active_dims=None, variances=1, lengthscales=1, order=1,
                 normalization=True, difference=True, num_lags=None, low_rank=False, num_components=50, rank_bound=None,
                 sparsity='sqrt', name=None):
        self.input_dim = int(input_dim)
        self.active_dims = self._validate_active_dims(input_dim, active_dims)
        self.name = name
        self.num_features = num_features
        self.num_levels = num_levels
        self.len_examples = self._validate_number_of_features(input_dim, num_features)
        self.order = num_levels if (order <= 0 or order >= num_levels) else order                 # kernels.py:57
        if self.order != 1 and low_rank:
            raise NotImplementedError('Higher-order algorithms not compatible with low-rank mode (yet).')   # :59-60
        self.normalization = normalization
        self.difference = difference
        self.variances = self._validate_signature_param("variances", variances, num_levels + 1)   # :65
        self.sigma = 1.0                                                                          # :66
        self.low_rank, self.num_components, self.rank_bound, self.sparsity = self._validate_low_rank_params(
            low_rank, num_components, rank_bound, sparsity)
        if num_lags is None:
            self.num_lags = 0
        else:
            if not isinstance(num_lags, int) or num_lags < 0:
                raise ValueError('The variable num_lags most be a nonnegative integer or None.')  # :74-75
            self.num_lags = int(num_lags)
            if num_lags > 0:
                self.lags = 0.1 * np.asarray(range(1, num_lags + 1), dtype=np.float64)           # :79
                gamma = 1. / np.asarray(range(1, self.num_lags + 2), dtype=np.float64)           # :80
                self.gamma = gamma / np.sum(gamma)                                               # :81
        if lengthscales is not None:
            self.lengthscales = self._validate_signature_param("lengthscales", lengthscales, self.num_features)  # :84-86
        else:
            self.lengthscales = None
        self._base_params = (0.0, 0.0)
        self.rng = np.random.default_rng()   # low-rank mode: source of landmarks and projections (the reference uses TF's global RNG)
        self.device_draw = True              # ... drawn on the device for CUDA tensors (gpsig_lr_draw), on the host for host arrays

    # ---- validators (kernels.py:94-133) ------------------------------------------------------
    @staticmethod
    def _validate_active_dims(input_dim, active_dims):
        if active_dims is None:
            return slice(int(input_dim))
        if isinstance(active_dims, slice):
            return active_dims
        return np.asarray(active_dims, dtype=np.int64)

    def _validate_number_of_features(self, input_dim, num_features):
        if input_dim % num_features == 0:
            return int(input_dim / num_features)
        raise ValueError("The arguments num_features and input_dim are not consistent.")

    def _validate_low_rank_params(self, low_rank, num_components, rank_bound, sparsity):
        if low_rank is not None and low_rank == True:  # noqa: E712  (as the reference)
            if not type(low_rank) == bool:
                raise ValueError("Unknown low-rank argument: %s. It should be True of False." % low_rank)
            if sparsity not in ['log', 'sqrt', 'lin']:
                raise ValueError("Unknown sparsity argument %s. Possible values are 'sqrt', 'log', 'lin'" % sparsity)
            if rank_bound is not None and rank_bound <= 0:
                raise ValueError("The rank-bound in the low-rank algorithm must be either None or a positiv integer.")
            if num_components is None or num_components <= 0:
                raise ValueError("The number of components in the kernel approximation must be a positive integer.")
            if rank_bound is None:
                rank_bound = num_components
        else:
            low_rank = False
        return low_rank, num_components, rank_bound, sparsity

    def _validate_signature_param(self, name, value, length):
        value = value * np.ones(length, dtype=np.float64)
        correct_shape = () if length == 1 else (length,)
        if np.asarray(value).squeeze().shape != correct_shape:
            raise ValueError("shape of parameter {} is not what is expected ({})".format(name, length))
        return value

    # ---- plumbing ------------------------------------------------------------------------------
    def _params(self, keep, dtype_id=_lib.F64):
        if self._base is None:
            raise NotImplementedError("SignatureKernel is abstract: use SignatureLinear, SignatureRBF, ...")
        p = _lib.Params()
        p.base_kernel = _lib.BASE[self._base]
        p.dtype = dtype_id
        p.num_features, p.num_levels, p.order = int(self.num_features), int(self.num_levels), int(self.order)
        p.difference, p.normalization, p.num_lags = int(bool(self.difference)), int(bool(self.normalization)), int(self.num_lags)
        p.sigma, p.jitter = float(self.sigma), JITTER
        bp = self._current_base_params()
        for k in range(4):
            p.base_params[k] = float(bp[k]) if k < len(bp) else 0.0

        def host(v):
            a = np.ascontiguousarray(v, dtype=np.float64)
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_double))
        p.variances = host(np.asarray(self.variances).reshape(-1))
        p.lengthscales = host(np.asarray(self.lengthscales).reshape(-1)) if self.lengthscales is not None else None
        if self.num_lags > 0:
            p.lags, p.gamma = host(self.lags), host(self.gamma)
        return p

    def _current_base_params(self):
        return self._base_params

    def _slice(self, X, X2=None):
        """GPflow ``Kernel._slice``: keep the active columns."""
        def one(A):
            if A is None:
                return None
            if isinstance(self.active_dims, slice):
                return A[..., self.active_dims]
            idx = torch.as_tensor(self.active_dims, device=A.device) if _is_torch(A) else self.active_dims
            return A[..., idx]
        return one(X), one(X2)

    def _seq_dims(self, X):
        n, width = _shape(X)[0], int(np.prod(_shape(X)[1:]))
        if width % self.num_features != 0:
            raise ValueError("input width %d is not a multiple of num_features=%d" % (width, self.num_features))
        return n, width // self.num_features

    def _tens_dims(self, Z, increments):
        lt = self.num_levels * (self.num_levels + 1) // 2
        shp = _shape(Z)
        d_eff = self.num_features * (self.num_lags + 1)
        want = (lt, shp[1], 2, d_eff) if increments else (lt, shp[1], d_eff)
        if len(shp) != len(want) or shp != want:
            raise ValueError("inducing tensors have shape %s, expected %s" % (shp, want))
        return shp[1]

    # ---- kernel evaluations ----------------------------------------------------------------------
    def graphed(self, method, *tensors, **kwargs):
        """HIP graph of one evaluation (no reference analogue): an evaluation is 5-15 short kernels, and a recorded graph
        replays them with one launch (include/gpsig_hip.h: gpsig_graph_begin).  It saves host time per call (12 us against
        25-40 us); the device time of a small evaluation is the serial lattice sweep of a pair and stays what it was.

            g = kern.graphed("K", X)            # X: contiguous CUDA tensor; evaluates once, then records
            X.copy_(X_next); K = g.replay()     # same shapes, new contents; K is g.out, overwritten by every replay

        For the methods that are library calls end to end (K, Kdiag, K_tens, K_tens_vs_seq, exact mode); hyper-parameters are
        baked in -- record again after changing them."""
        if torch is None or not tensors or not all(t is None or (_is_torch(t) and t.is_cuda and t.is_contiguous()) for t in tensors):
            raise ValueError("graphed() takes contiguous CUDA tensors")
        if self.low_rank:
            raise NotImplementedError("low-rank evaluations draw random objects on the host and cannot be recorded")
        # nothing torch allocates may end up inside the capture (its memory is recycled after the call, a replay would write into
        # whatever lives there then): the tensors must be what the library reads as they are -- one dtype, all columns active
        given = [t for t in tensors if t is not None]
        if len({t.dtype for t in given}) != 1 or given[0].dtype not in (torch.float32, torch.float64):
            raise ValueError("graphed() takes tensors of one dtype, float32 or float64 (a conversion would be recorded with a temporary)")
        ad = self.active_dims
        if not (ad is None or (isinstance(ad, slice) and ad == slice(None)) or (isinstance(ad, slice) and ad.start in (None, 0)
                and ad.step in (None, 1) and (ad.stop is None or ad.stop >= self.input_dim))):
            raise ValueError("graphed() needs the default active_dims (a column gather would be recorded with a temporary)")
        fn = getattr(self, method)
        dev = next(t for t in tensors if t is not None).device
        side = torch.cuda.Stream(dev)            # the default stream cannot be captured
        side.wait_stream(torch.cuda.current_stream(dev))
        self._graph_recording = True             # _f32_upcast refuses instead of converting (its temporaries must not be recorded)
        try:
            with torch.cuda.stream(side):
                fn(*tensors, **kwargs)           # scratch buffers, task lists and level weights in place
                ctx = _lib.context(dev.index or 0, side.cuda_stream)
                _lib.hold(dev.index or 0, side.cuda_stream)
                try:
                    with ctx.graph() as g:
                        out = fn(*tensors, **kwargs)
                except Exception:
                    _lib.release(dev.index or 0, side.cuda_stream)
                    raise
        finally:
            self._graph_recording = False
        torch.cuda.current_stream(dev).wait_stream(side)
        return GraphedCall(g, tensors, out, side)

    @_f32_upcast
    def K(self, X, X2=None, presliced=False, return_levels=False, presliced_X=False, presliced_X2=False, lr_state=None):
        """Reference: kernels.py:401-476.  (N1, N2) or (M+1, N1, N2).  lr_state: low-rank mode only, the random
        objects to use (default: drawn afresh, as the reference does)."""
        if presliced:
            presliced_X = presliced_X2 = True
        if not presliced_X:
            X, _ = self._slice(X, None)
        if not presliced_X2 and X2 is not None:
            X2, _ = self._slice(X2, None)
        if self.low_rank:
            return self._K_lr(X, X2, return_levels, lr_state)
        L_ = _Launch(X, X2)
        n1, l1 = self._seq_dims(X)
        n2, l2 = self._seq_dims(X2) if X2 is not None else (n1, l1)
        p = self._params(L_.keep, L_.dtype_id)
        out, optr = L_.out((self.num_levels + 1, n1, n2) if return_levels else (n1, n2))
        L_.ctx.call("gpsig_kernel_K", p, L_.inp(X), L_.inp(X2), n1, n2, l1, l2, int(bool(return_levels)), optr)
        return out

    @_f32_upcast
    def Kdiag(self, X, presliced=False, return_levels=False, lr_state=None):
        """Reference: kernels.py:479-510.  (N,) or (M+1, N)."""
        if not presliced:
            X, _ = self._slice(X, None)
        if self.low_rank and not self.normalization:
            L_ = _launch_f64(X)
            st = lr_state or self.draw_low_rank(X=X, _implicit=True)
            p = self._params(L_.keep)
            lr = st.as_c(L_.keep)
            Phi, pp, n = self._lr_features(L_, p, lr, X)
            out, optr = L_.out((self.num_levels + 1, n) if return_levels else (n,))
            L_.ctx.call("gpsig_lr_kernel_diag", p, lr, pp, n, int(bool(return_levels)), optr)
            return out
        L_ = _Launch(X)
        n, l = self._seq_dims(X)
        p = self._params(L_.keep, L_.dtype_id)
        out, optr = L_.out((self.num_levels + 1, n) if return_levels else (n,))
        L_.ctx.call("gpsig_kernel_Kdiag", p, L_.inp(X), n, l, int(bool(return_levels)), optr)
        return out

    @_f32_upcast
    def K_tens(self, Z, return_levels=False, increments=False, lr_state=None):
        """Reference: kernels.py:513-536.  (T, T) or (M+1, T, T); never normalised."""
        if self.low_rank:
            L_ = _launch_f64(Z)
            st = lr_state or self.draw_low_rank(Z=Z, increments=increments, _implicit=True)
            p = self._params(L_.keep)
            lr = st.as_c(L_.keep)
            Phi, pp, t = self._lr_features(L_, p, lr, Z, tensors=True, increments=increments)
            out, optr = L_.out((self.num_levels + 1, t, t) if return_levels else (t, t))
            L_.ctx.call("gpsig_lr_kernel", p, lr, pp, None, t, t, 0, 0, int(bool(return_levels)), optr)
            return out
        L_ = _Launch(Z)
        t = self._tens_dims(Z, increments)
        p = self._params(L_.keep, L_.dtype_id)
        out, optr = L_.out((self.num_levels + 1, t, t) if return_levels else (t, t))
        L_.ctx.call("gpsig_kernel_K_tens", p, L_.inp(Z), t, int(bool(increments)), int(bool(return_levels)), optr)
        return out

    @_f32_upcast
    def K_tens_vs_seq(self, Z, X, return_levels=False, increments=False, presliced=False, lr_state=None):
        """Reference: kernels.py:539-588.  (T, N) or (M+1, T, N); normalised on the sequence axis only."""
        if not presliced:
            X, _ = self._slice(X, None)
        if self.low_rank:
            L_ = _launch_f64(Z, X)
            st = lr_state or self.draw_low_rank(X=X, Z=Z, increments=increments, _implicit=True)
            p = self._params(L_.keep)
            lr = st.as_c(L_.keep)
            PZ, pz, t = self._lr_features(L_, p, lr, Z, tensors=True, increments=increments)
            PX, px, n = self._lr_features(L_, p, lr, X)
            out, optr = L_.out((self.num_levels + 1, t, n) if return_levels else (t, n))
            L_.ctx.call("gpsig_lr_kernel", p, lr, pz, px, t, n, 0, int(bool(self.normalization)), int(bool(return_levels)), optr)
            return out
        L_ = _Launch(Z, X)
        t = self._tens_dims(Z, increments)
        n, l = self._seq_dims(X)
        p = self._params(L_.keep, L_.dtype_id)
        out, optr = L_.out((self.num_levels + 1, t, n) if return_levels else (t, n))
        L_.ctx.call("gpsig_kernel_K_tens_vs_seq", p, L_.inp(Z), L_.inp(X), t, n, l, int(bool(increments)),
                    int(bool(return_levels)), optr)
        return out

    @_f32_upcast
    def K_tens_n_seq_covs(self, Z, X, full_X_cov=False, return_levels=False, increments=False, presliced=False):
        """Reference: kernels.py:591-671.  Returns (Kzz, Kzx, Kxx); Kxx is the diagonal unless full_X_cov."""
        if not presliced:
            X, _ = self._slice(X, None)
        if self.low_rank:
            # one shared draw of landmarks / projections for all three matrices (kernels.py:613-621)
            st = self.draw_low_rank(X=X, Z=Z, increments=increments)
            L_ = _launch_f64(Z, X)
            p = self._params(L_.keep)
            lr = st.as_c(L_.keep)
            lv, nrm = int(bool(return_levels)), int(bool(self.normalization))
            m1 = (self.num_levels + 1,) if return_levels else ()
            # the two factor matrices once (kernels.py:613-621), then the three products of :623-661
            PZ, pz, t = self._lr_features(L_, p, lr, Z, tensors=True, increments=increments)
            PX, px, n = self._lr_features(L_, p, lr, X)
            Kzz, ozz = L_.out(m1 + (t, t))
            L_.ctx.call("gpsig_lr_kernel", p, lr, pz, None, t, t, 0, 0, lv, ozz)
            Kzx, ozx = L_.out(m1 + (t, n))
            L_.ctx.call("gpsig_lr_kernel", p, lr, pz, px, t, n, 0, nrm, lv, ozx)      # :638 == :581: divided by the X side's norms only
            if full_X_cov:
                Kxx, oxx = L_.out(m1 + (n, n))
                L_.ctx.call("gpsig_lr_kernel", p, lr, px, None, n, n, nrm, nrm, lv, oxx)
            elif self.normalization:
                Kxx = self.Kdiag(X, presliced=True, return_levels=return_levels, lr_state=st)     # sigma * variances: no features needed
            else:
                Kxx, oxx = L_.out(m1 + (n,))
                L_.ctx.call("gpsig_lr_kernel_diag", p, lr, px, n, lv, oxx)
            return Kzz, Kzx, Kxx
        L_ = _Launch(Z, X)
        t = self._tens_dims(Z, increments)
        n, l = self._seq_dims(X)
        p = self._params(L_.keep, L_.dtype_id)
        lv = (self.num_levels + 1,) if return_levels else ()
        Kzz, pzz = L_.out(lv + (t, t))
        Kzx, pzx = L_.out(lv + (t, n))
        Kxx, pxx = L_.out(lv + ((n, n) if full_X_cov else (n,)))
        L_.ctx.call("gpsig_kernel_K_tens_n_seq_covs", p, L_.inp(Z), L_.inp(X), t, n, l, int(bool(increments)),
                    int(bool(full_X_cov)), int(bool(return_levels)), pzz, pzx, pxx)
        return Kzz, Kzx, Kxx

    @_f32_upcast
    def K_seq_n_seq_covs(self, X, X2, full_X2_cov=False, return_levels=False, presliced=False, lr_state=None):
        """Reference: kernels.py:674-761 (X = inducing sequences, X2 = data).  Returns (Kxx, Kxx2, Kx2x2).
        The double division of Kxx2 by the X-side diagonal in the diagonal-only branch (:713 + :750) is
        reproduced; the undefined names of :723-728 are read as the evident mirror of :709-712."""
        if not presliced:
            X2, _ = self._slice(X2, None)
        if self.low_rank:
            return self._K_seq_n_seq_covs_lr(X, X2, full_X2_cov, return_levels, lr_state)
        L_ = _Launch(X, X2)
        n1, l1 = self._seq_dims(X)
        n2, l2 = self._seq_dims(X2)
        p = self._params(L_.keep, L_.dtype_id)
        lv = (self.num_levels + 1,) if return_levels else ()
        Kxx, p11 = L_.out(lv + (n1, n1))
        Kxx2, p12 = L_.out(lv + (n1, n2))
        Kx2x2, p22 = L_.out(lv + ((n2, n2) if full_X2_cov else (n2,)))
        L_.ctx.call("gpsig_kernel_K_seq_n_seq_covs", p, L_.inp(X), L_.inp(X2), n1, n2, l1, l2, int(bool(full_X2_cov)),
                    int(bool(return_levels)), p11, p12, p22)
        return Kxx, Kxx2, Kx2x2

    # ---- low-rank mode (kernels.py:239-311 and the low_rank branches of K / Kdiag / K_tens / K_tens_vs_seq) -----
    def _scaled_tensor_points(self, Z, increments):
        """kernels.py:367-398 on the host (inducing tensors are small): flat scaled components (rows, d')."""
        Z = np.asarray(Z.detach().cpu().numpy() if _is_torch(Z) else Z, dtype=np.float64)
        d_eff = self.num_features * (self.num_lags + 1)
        Zf = Z.reshape(-1, self.num_lags + 1, self.num_features).copy()
        if self.lengthscales is not None:
            Zf = Zf / np.asarray(self.lengthscales)[None, None, :]
            if self.num_lags > 0:
                Zf = Zf * np.asarray(self.gamma)[None, :, None]
        return Zf.reshape(-1, d_eff)

    def draw_low_rank(self, X=None, X2=None, Z=None, increments=False, _implicit=False):
        """Draw the landmarks (uniformly, without replacement, from the scaled points of every given argument:
        kernels.py:444-446, :562-563), whiten their Gram (low_rank_calculations.py:50-57) and draw one projection
        per level (low_rank_calculations.py:76-193).  Returns a LowRankState that can be passed to K(..., lr_state=).
        A state returned to the caller is the caller's: later draws never write into it, for host arrays and CUDA tensors alike
        (only the per-evaluation draws of K / Kdiag / K_tens / ... with lr_state=None reuse one device block of the kernel object)."""
        L_ = _launch_f64(X, X2, Z)
        if L_.device_mode and self.device_draw:
            return self._draw_low_rank_on_device(L_, X, X2, Z, increments, reuse=_implicit)
        L_ = _launch_f64(X, X2)
        p = self._params(L_.keep, _lib.F64)
        total = 0
        seqs = []
        for A in (X, X2):
            if A is not None:
                n, l = self._seq_dims(A)
                seqs.append((A, n, l, total))
                total += n * l
        ztot = 0
        if Z is not None:
            zp = self._scaled_tensor_points(Z, increments)
            ztot = zp.shape[0]
        c = int(self.num_components)
        if c > total + ztot:
            raise ValueError("num_components exceeds the number of available points")
        pick = np.sort(self.rng.choice(total + ztot, size=c, replace=False, shuffle=False))      # c of all points, without shuffling them all
        d_eff = self.num_features * (self.num_lags + 1)
        parts = []
        if Z is not None:
            parts.append(zp[pick[pick < ztot]])
        for A, n, l, off in seqs:
            sel = pick[(pick >= ztot + off) & (pick < ztot + off + n * l)] - ztot - off
            out = np.empty((sel.shape[0], d_eff))
            if sel.shape[0]:
                idx = np.ascontiguousarray(sel, dtype=np.int64)
                L_.ctx.call("gpsig_lr_gather_points", p, L_.inp(A), n, l, idx.ctypes.data_as(C.POINTER(C.c_int64)), idx.shape[0],
                            out.ctypes.data_as(C.POINTER(C.c_double)))
            parts.append(out)
        S = np.ascontiguousarray(np.concatenate(parts, axis=0))
        jd = np.ascontiguousarray(JITTER * self.rng.random(c))                    # low_rank_calculations.py:52
        sk = _lr.draw_level_sketches(self.rng, self.num_levels, c, int(self.rank_bound), self.sparsity)
        return self.low_rank_state(S, jd, sk, ctx=L_.ctx)

    def _draw_low_rank_on_device(self, L_, X, X2, Z, increments, reuse=False):
        """gpsig_lr_draw: landmark choice, gather, whitening (Jacobi eigendecomposition) and the projections of every level on the
        tensors' own stream, seeded from ``self.rng``; nothing waits for the host.  Returns a DeviceLowRankState.
        reuse: draw into the kernel object's own state (the implicit draw of an evaluation that was given no lr_state: nobody else
        holds it, and a fresh device block per evaluation would be a hipMalloc -- a device-wide synchronisation -- each);
        otherwise a fresh state that belongs to the caller."""
        p = self._params(L_.keep, _lib.F64)
        n1, l1 = self._seq_dims(X) if X is not None else (0, 1)
        n2, l2 = self._seq_dims(X2) if X2 is not None else (0, 1)
        t = self._tens_dims(Z, increments) if Z is not None else 0
        st = getattr(self, "_device_lr_state", None) if reuse else None
        if st is None or st.ctx is not L_.ctx or getattr(st.ctx, "_h", None) is None or (st.num_components, st.rank_bound, st.num_sketches) != (
                int(self.num_components), int(self.rank_bound), self.num_levels - 1):
            st = DeviceLowRankState(L_.ctx, self.num_components, self.rank_bound, self.num_levels - 1)
        seed = int(self.rng.integers(0, 2 ** 63 - 1))
        L_.ctx.call("gpsig_lr_draw", p, int(self.num_components), int(self.rank_bound), DeviceLowRankState.SPARSITY[self.sparsity], C.c_uint64(seed),
                    L_.inp(X), n1, l1, L_.inp(X2), n2, l2, L_.inp(Z), t, int(bool(increments)), C.byref(st._h))
        st.keep = L_.keep            # the (converted) inputs stay alive until the draw has read them
        if reuse:
            self._device_lr_state = st
        return st

    def low_rank_state(self, landmarks, jitter_diag, sketches, ctx=None):
        """The LowRankState of GIVEN random objects: landmarks (c, d') -- scaled points --, the jitter draw (c,) and one sketch
        per level >= 2 (objects with k1, k2, r, colptr, i1, i2, val); the whitening is computed here, on the device
        (low_rank_calculations.py:50-57, :60: landmark Gram + jitter, rocSOLVER dsyevd, U / sqrt(S + jitter)).
        ctx: the library context to whiten on (default: device 0's); its pointer mode is left alone -- gpsig_lr_whitening takes host
        pointers in either mode, and the caller's evaluation may be half way through a sequence of device-pointer calls on it."""
        S = np.ascontiguousarray(landmarks, dtype=np.float64)
        jd = np.ascontiguousarray(jitter_diag, dtype=np.float64)
        c, d_eff = S.shape
        keep = []
        p = self._params(keep, _lib.F64)
        Wh, ev = np.empty((c, c)), np.empty(c)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))                    # noqa: E731
        (ctx or _lib.context(0, 0)).call("gpsig_lr_whitening", p, dp(S), c, d_eff, dp(jd), dp(Wh), dp(ev))
        sk = [s_ if isinstance(s_, _lr.Sketch) else _lr.Sketch(s_.k1, s_.k2, s_.r, s_.colptr, s_.i1, s_.i2, s_.val) for s_ in sketches]
        rb = sk[0].r if sk else int(self.rank_bound)
        return LowRankState(S, Wh, sk, rb, jitter_diag=jd, eigenvalues=ev)

    def _lr_features(self, L_, p, lr, A, tensors=False, increments=False):
        F = 1 + lr.num_components + (self.num_levels - 1) * lr.rank_bound
        if tensors:
            t = self._tens_dims(A, increments)
            Phi, pp = L_.out((t, F))
            L_.ctx.call("gpsig_lr_tens_features", p, lr, L_.inp(A), t, int(bool(increments)), pp)
            return Phi, pp, t
        n, l = self._seq_dims(A)
        Phi, pp = L_.out((n, F))
        L_.ctx.call("gpsig_lr_seq_features", p, lr, L_.inp(A), n, l, pp)
        return Phi, pp, n

    def _K_lr(self, X, X2, return_levels, lr_state):
        L_ = _launch_f64(X, X2)
        st = lr_state or self.draw_low_rank(X=X, X2=X2, _implicit=True)
        p = self._params(L_.keep)
        lr = st.as_c(L_.keep)
        PA, pa, n1 = self._lr_features(L_, p, lr, X)
        if X2 is None:
            PB, pb, n2 = None, None, n1
        else:
            PB, pb, n2 = self._lr_features(L_, p, lr, X2)
        out, optr = L_.out((self.num_levels + 1, n1, n2) if return_levels else (n1, n2))
        nrm = int(bool(self.normalization))
        L_.ctx.call("gpsig_lr_kernel", p, lr, pa, pb, n1, n2, nrm, nrm, int(bool(return_levels)), optr)
        return out

    def _K_seq_n_seq_covs_lr(self, X, X2, full_X2_cov, return_levels, lr_state):
        """kernels.py:696-761, low-rank branch: level Grams of the factor matrices (HIP: features + fp64-MFMA GEMMs), then the
        normalisation / weighting of :706-761 as elementwise torch ops on the device."""
        L_ = _launch_f64(X, X2)
        st = lr_state or self.draw_low_rank(X=X, X2=X2, _implicit=True)
        p = self._params(L_.keep)
        ones = np.ones(self.num_levels + 1)
        L_.keep.append(ones)
        p.sigma, p.variances = 1.0, ones.ctypes.data_as(C.POINTER(C.c_double))      # raw level Grams; weights are applied below
        lr = st.as_c(L_.keep)
        P1, pp1, n1 = self._lr_features(L_, p, lr, X)
        P2, pp2, n2 = self._lr_features(L_, p, lr, X2)
        M1 = self.num_levels + 1
        Kxx, o11 = L_.out((M1, n1, n1))
        Kxx2, o12 = L_.out((M1, n1, n2))
        L_.ctx.call("gpsig_lr_kernel", p, lr, pp1, None, n1, n1, 0, 0, 1, o11)                     # :702
        L_.ctx.call("gpsig_lr_kernel", p, lr, pp1, pp2, n1, n2, 0, 0, 1, o12)                      # :703
        if full_X2_cov:
            K22, o22 = L_.out((M1, n2, n2))
            L_.ctx.call("gpsig_lr_kernel", p, lr, pp2, None, n2, n2, 0, 0, 1, o22)                 # :718
        else:
            K22, o22 = L_.out((M1, n2))
            L_.ctx.call("gpsig_lr_kernel_diag", p, lr, pp2, n2, 1, o22)                            # :740
        if not L_.device_mode:
            L_.ctx.sync()
        dev = L_.dev if L_.device_mode else torch.device("cuda", 0)
        t = lambda a: a if _is_torch(a) else torch.as_tensor(a, device=dev)
        Kxx, Kxx2, K22 = t(Kxx), t(Kxx2), t(K22)
        w = torch.as_tensor(float(self.sigma) * np.asarray(self.variances, dtype=np.float64), device=dev)
        if self.normalization:
            Kxx = Kxx + JITTER * torch.eye(n1, dtype=Kxx.dtype, device=dev)[None]                   # :709
            dsq = torch.sqrt(torch.diagonal(Kxx, dim1=1, dim2=2))
            Kxx = Kxx / (dsq[:, :, None] * dsq[:, None, :])
            Kxx2 = Kxx2 / dsq[:, :, None]                                                           # :713
        if full_X2_cov:
            if self.normalization:                                                                  # :723-728 (intent; undefined names there)
                K22 = K22 + JITTER * torch.eye(n2, dtype=K22.dtype, device=dev)[None]
                d2 = torch.sqrt(torch.diagonal(K22, dim1=1, dim2=2))
                Kxx2 = Kxx2 / d2[:, None, :]
                K22 = K22 / (d2[:, :, None] * d2[:, None, :])
            K22 = K22 * w[:, None, None]
        else:
            if self.normalization:
                d2 = torch.sqrt(K22 + JITTER)                                                       # :746-748
                Kxx2 = Kxx2 / (dsq[:, :, None] * d2[:, None, :])                                    # :750 (second division by dsq: reference quirk)
                K22 = w[:, None].expand(-1, n2).clone()                                             # :751
            else:
                K22 = K22 * w[:, None]
        Kxx, Kxx2 = Kxx * w[:, None, None], Kxx2 * w[:, None, None]
        if not return_levels:
            Kxx, Kxx2, K22 = Kxx.sum(0), Kxx2.sum(0), K22.sum(0)
        if not L_.device_mode:
            return Kxx.cpu().numpy(), Kxx2.cpu().numpy(), K22.cpu().numpy()
        return Kxx, Kxx2, K22

    # ---- the signature_algs.py layer: unnormalised level tensors -----------------------------------
    def _K_seq(self, X, X2=None):
        """Reference: kernels.py:208-237 on already scaled (N, L, d') sequences -> (M+1, N1, N2)."""
        L_ = _Launch(X, X2)
        n1, l1 = _shape(X)[0], _shape(X)[1]
        n2, l2 = (_shape(X2)[0], _shape(X2)[1]) if X2 is not None else (n1, l1)
        p = self._params(L_.keep, L_.dtype_id)
        out, optr = L_.out((self.num_levels + 1, n1, n2))
        L_.ctx.call("gpsig_seq_gram_levels", p, L_.inp(X), L_.inp(X2), n1, n2, l1, l2, optr)
        return out

    def _K_seq_diag(self, X):
        """Reference: kernels.py:188-205 -> (M+1, N)."""
        L_ = _Launch(X)
        n, l = _shape(X)[0], _shape(X)[1]
        p = self._params(L_.keep, L_.dtype_id)
        out, optr = L_.out((self.num_levels + 1, n))
        L_.ctx.call("gpsig_seq_diag_levels", p, L_.inp(X), n, l, optr)
        return out

    def _K_tens(self, Z, increments=False):
        """Reference: kernels.py:263-283 on already scaled tensors -> (M+1, T, T)."""
        L_ = _Launch(Z)
        t = _shape(Z)[1]
        p = self._params(L_.keep, L_.dtype_id)
        out, optr = L_.out((self.num_levels + 1, t, t))
        L_.ctx.call("gpsig_tens_gram_levels", p, L_.inp(Z), t, int(bool(increments)), optr)
        return out

    def _K_tens_vs_seq(self, Z, X, increments=False):
        """Reference: kernels.py:313-340 on already scaled inputs -> (M+1, T, N)."""
        L_ = _Launch(Z, X)
        t, n, l = _shape(Z)[1], _shape(X)[0], _shape(X)[1]
        p = self._params(L_.keep, L_.dtype_id)
        out, optr = L_.out((self.num_levels + 1, t, n))
        L_.ctx.call("gpsig_tens_vs_seq_levels", p, L_.inp(Z), L_.inp(X), t, n, l, int(bool(increments)), optr)
        return out

    # ---- numpy-facing wrappers (kernels.py:141-186; GPflow autoflow in the reference) --------------------
    def compute_K(self, X, Y):
        return self.K(X, Y)

    def compute_K_symm(self, X):
        return self.K(X)

    def compute_base_kern_symm(self, X):
        """Reference: kernels.py:150-157.  The static kernel on every pair of (scaled, lagged) observations of X, (N, N, L, L)."""
        X, _ = self._slice(X, None)
        L_ = _Launch(X)
        if L_.f32:
            raise NotImplementedError("compute_base_kern_symm is built for float64 only")
        n, l = self._seq_dims(X)
        p = self._params(L_.keep, _lib.F64)
        d_eff = self.num_features * (self.num_lags + 1)
        pts = np.empty((n * l, d_eff))
        if n * l:
            idx = np.arange(n * l, dtype=np.int64)
            L_.ctx.call("gpsig_lr_gather_points", p, L_.inp(X), n, l, idx.ctypes.data_as(C.POINTER(C.c_int64)), n * l,
                        pts.ctypes.data_as(C.POINTER(C.c_double)))                                     # :152-154 (scaling, lags)
        W = np.empty((n * l, n * l))
        L_.ctx.call("gpsig_base_kernel_matrix", p, pts.ctypes.data_as(C.POINTER(C.c_double)), pts.ctypes.data_as(C.POINTER(C.c_double)),
                    n * l, n * l, d_eff, W.ctypes.data_as(C.POINTER(C.c_double)))                        # :155
        out = W.reshape(n, l, n, l).transpose(0, 2, 1, 3)                                               # :156-157
        return torch.as_tensor(np.ascontiguousarray(out), device=L_.dev) if L_.device_mode else out

    def compute_K_level_diags(self, X):
        return self.Kdiag(X, return_levels=True)

    def compute_K_levels(self, X, X2):
        return self.K(X, X2, return_levels=True)

    def compute_Kdiag(self, X):
        return self.Kdiag(X)

    def compute_K_tens(self, Z):
        return self.K_tens(Z, return_levels=False)

    def compute_K_tens_vs_seq(self, Z, X):
        return self.K_tens_vs_seq(Z, X, return_levels=False)

    def compute_K_incr_tens(self, Z):
        return self.K_tens(Z, increments=True, return_levels=False)

    def compute_K_incr_tens_vs_seq(self, Z, X):
        return self.K_tens_vs_seq(Z, X, increments=True, return_levels=False)


class SignatureLinear(SignatureKernel):
    """Identity state-space embedding (kernels.py:786-806)."""
    _base = "linear"


class SignatureCosine(SignatureKernel):
    """Cosine similarity as state-space kernel (kernels.py:808-828)."""
    _base = "cosine"


class SignaturePoly(SignatureKernel):
    """Polynomial state-space kernel (x.y + gamma)^degree (kernels.py:831-848)."""
    _base = "poly"

    def __init__(self, input_dim, num_features, num_levels, gamma=1, degree=3, **kwargs):
        SignatureKernel.__init__(self, input_dim, num_features, num_levels, **kwargs)
        if self.num_lags > 0:
            # the reference stores the offset in self.gamma, overwriting the lag weights of kernels.py:82 (:837)
            raise NotImplementedError("SignaturePoly with num_lags > 0: the reference overwrites the lag weights (kernels.py:82 vs :837)")
        self.gamma = float(gamma)
        self.degree = float(degree)

    def _current_base_params(self):
        return (float(self.gamma), float(self.degree))

    def _set_base_p0(self, value):
        self.gamma = float(value)


class SignatureRBF(SignatureKernel):
    """Gaussian state-space kernel exp(-|x-y|^2/2) on the scaled inputs (kernels.py:850-864)."""
    _base = "rbf"


SignatureGauss = SignatureRBF


class SignatureMix(SignatureKernel):
    """mixing * RBF + (1 - mixing) * linear (kernels.py:870-892)."""
    _base = "mix"

    def __init__(self, input_dim, num_features, num_levels, **kwargs):
        SignatureKernel.__init__(self, input_dim, num_features, num_levels, **kwargs)
        self.mixing = 0.5

    def _current_base_params(self):
        return (float(self.mixing), 0.0)

    def _set_base_p0(self, value):
        self.mixing = float(value)


class SignatureSpectral(SignatureKernel):
    """Spectral-mixture state-space kernels (kernels.py:894-942):
    kappa(x, y) = sum_q alpha_q * E_q(x - y) * cos(2 pi <omega_q, x - y>), E_q Gaussian ('gauss' / 'rbf'), exponential ('exp')
    or, for 'mixed', Gaussian for the first floor(Q/2) components and exponential for the rest (the reference's 'mixed' branch
    references an undefined name and has a sign slip, :932-936; the evident intent is built).  As in the reference there is
    no lengthscale scaling (:907) -- gamma (Q, num_features) plays that role."""
    _base = "spectral"
    _FAMILIES = {'exp': 1, 'exponential': 1, 'gauss': 0, 'gaussian': 0, 'rbf': 0, 'mixed': 2, 'mix': 2}

    def __init__(self, input_dim, num_features, num_levels, family='gauss', Q=5, **kwargs):
        kwargs.pop("lengthscales", None)
        SignatureKernel.__init__(self, input_dim, num_features, num_levels, lengthscales=None, **kwargs)     # :907
        if family not in self._FAMILIES:
            raise ValueError("Unrecognized spectral family name.")                                           # :916
        if self.num_lags > 0:
            # the reference stores its (Q, d) scale matrix in self.gamma, overwriting the lag weights of kernels.py:82 (:913)
            raise NotImplementedError("SignatureSpectral with num_lags > 0: the reference overwrites the lag weights (kernels.py:82 vs :913)")
        if self.low_rank:
            raise NotImplementedError("SignatureSpectral in low-rank mode is not built")
        self.family = {0: 'rbf', 1: 'exp', 2: 'mixed'}[self._FAMILIES[family]]                               # :909-914
        self.Q = int(Q)
        self.alpha = np.exp(np.random.randn(self.Q))                                                         # :919
        self.omega = np.exp(np.random.randn(self.Q, self.num_features))                                      # :920
        self.gamma = np.exp(np.random.randn(self.Q, self.num_features))                                      # :921

    def _current_base_params(self):
        return (float(self.Q), float(self._FAMILIES[self.family]))

    def _params(self, keep, dtype_id=_lib.F64):
        p = SignatureKernel._params(self, keep, dtype_id)
        tab = np.ascontiguousarray(np.concatenate([np.asarray(self.alpha, dtype=np.float64).reshape(-1),
                                                   np.asarray(self.omega, dtype=np.float64).reshape(-1),
                                                   np.asarray(self.gamma, dtype=np.float64).reshape(-1)]))
        if tab.shape[0] != self.Q * (1 + 2 * self.num_features):
            raise ValueError("alpha, omega, gamma must have shapes (Q,), (Q, num_features), (Q, num_features)")
        keep.append(tab)
        p.base_table = tab.ctypes.data_as(C.POINTER(C.c_double))
        p.base_table_len = tab.shape[0]
        p.lags, p.gamma = None, None
        return p


class SignatureMatern12(SignatureKernel):
    """exp(-r) (kernels.py:944-958)."""
    _base = "matern12"


SignatureLaplace = SignatureMatern12
SignatureExponential = SignatureMatern12


class SignatureMatern32(SignatureKernel):
    """(1 + sqrt(3) r) exp(-sqrt(3) r) (kernels.py:964-977)."""
    _base = "matern32"


class SignatureMatern52(SignatureKernel):
    """(1 + sqrt(5) r + 5/3 r^2) exp(-sqrt(5) r) (kernels.py:981-993)."""
    _base = "matern52"
