"""ctypes binding of libgpsig_hip.so (include/gpsig_hip.h).

The shared library is built in-tree by ``__graft_entry__.build()`` / ``make -C gpsig_amd/csrc``.
There is no CPU fallback: if the library or a HIP device is missing, importing works but the first
kernel evaluation raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPSIG_LIB") or os.path.join(_HERE, "lib", "libgpsig_hip.so")   # GPSIG_LIB: A/B builds

GPSIG_OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM = 0, -1, -2, -3, -4
PTR_HOST, PTR_DEVICE = 0, 1
F64, F32 = 0, 1
BASE = {"linear": 0, "rbf": 1, "cosine": 2, "poly": 3, "mix": 4, "matern12": 5, "matern32": 6, "matern52": 7, "spectral": 8}


class Params(C.Structure):
    """struct gpsig_params (include/gpsig_hip.h)."""
    _fields_ = [
        ("base_kernel", C.c_int32), ("dtype", C.c_int32), ("num_features", C.c_int32), ("num_levels", C.c_int32),
        ("order", C.c_int32), ("difference", C.c_int32), ("normalization", C.c_int32), ("num_lags", C.c_int32),
        ("sigma", C.c_double), ("jitter", C.c_double), ("base_params", C.c_double * 4),
        ("variances", C.POINTER(C.c_double)), ("lengthscales", C.POINTER(C.c_double)),
        ("lags", C.POINTER(C.c_double)), ("gamma", C.POINTER(C.c_double)),
        ("base_table", C.POINTER(C.c_double)), ("base_table_len", C.c_int64),
    ]


class SketchC(C.Structure):
    """struct gpsig_sketch."""
    _fields_ = [("k1", C.c_int32), ("k2", C.c_int32), ("r", C.c_int32), ("nnz", C.c_int32),
                ("colptr", C.POINTER(C.c_int32)), ("i1", C.POINTER(C.c_int32)), ("i2", C.POINTER(C.c_int32)),
                ("val", C.POINTER(C.c_double))]


class LowRankC(C.Structure):
    """struct gpsig_lowrank."""
    _fields_ = [("num_components", C.c_int32), ("rank_bound", C.c_int32), ("num_sketches", C.c_int32),
                ("landmarks", C.POINTER(C.c_double)), ("whitening", C.POINTER(C.c_double)), ("sketches", C.POINTER(SketchC)),
                ("device_state", C.c_void_p)]


_P = C.POINTER(Params)
_LR = C.POINTER(LowRankC)
_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64

# name -> argtypes after (ctx, params); every symbol include/gpsig_hip.h declares must appear here or in _PLAIN
_KERNEL_FUNCS = {
    "gpsig_seq_gram_levels": [_vp, _vp, _i64, _i64, _i32, _i32, _vp],
    "gpsig_seq_diag_levels": [_vp, _i64, _i32, _vp],
    "gpsig_tens_gram_levels": [_vp, _i64, _i32, _vp],
    "gpsig_tens_vs_seq_levels": [_vp, _vp, _i64, _i64, _i32, _i32, _vp],
    "gpsig_lattice_levels": [_vp, _i64, _i32, _i32, _vp],
    "gpsig_lattice_levels_grad": [_vp, _i64, _i32, _i32, _vp, _vp],
    "gpsig_chain_levels": [_vp, _i64, _i32, _vp],
    "gpsig_chain_levels_grad": [_vp, _i64, _i32, _vp, _vp],
    "gpsig_tens_vs_seq_weighted": [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, C.POINTER(_i32)],
    "gpsig_tens_vs_seq_weighted_grad": [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_double)],
    "gpsig_kernel_K": [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp],
    "gpsig_seq_features": [_vp, _i64, _i32, _vp],
    "gpsig_seq_features_grad": [_vp, _i64, _i32, _vp, _vp, _vp],
    "gpsig_kernel_K_grad": [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, C.POINTER(_i32)],
    "gpsig_kernel_K_symm_rows": [_vp, _i64, _i32, _i64, _i64, _vp],
    "gpsig_kernel_K_symm_rows_compact": [_vp, _i64, _i32, _i64, _i64, _vp],
    "gpsig_kernel_Kdiag": [_vp, _i64, _i32, _i32, _vp],
    "gpsig_kernel_K_tens": [_vp, _i64, _i32, _i32, _vp],
    "gpsig_kernel_K_tens_vs_seq": [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp],
    "gpsig_kernel_K_tens_n_seq_covs": [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "gpsig_kernel_K_seq_n_seq_covs": [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "gpsig_lr_draw": [_i32, _i32, _i32, C.c_uint64, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, C.POINTER(_vp)],
    "gpsig_lr_gather_points": [_vp, _i64, _i32, C.POINTER(_i64), _i64, C.POINTER(C.c_double)],
    "gpsig_base_kernel_matrix": [C.POINTER(C.c_double), C.POINTER(C.c_double), _i64, _i64, _i32, C.POINTER(C.c_double)],
    "gpsig_lr_whitening": [C.POINTER(C.c_double), _i32, _i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "gpsig_lr_seq_features": [_LR, _vp, _i64, _i32, _vp],
    "gpsig_lr_tens_features": [_LR, _vp, _i64, _i32, _vp],
    "gpsig_lr_seq_features_dev": [_i32, _i32, _i32, C.POINTER(SketchC), _vp, _i64, _i32, _vp, _vp, _vp],
    "gpsig_lr_seq_features_grad": [_i32, _i32, _i32, C.POINTER(SketchC), _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_double)],
    "gpsig_lr_kernel": [_LR, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp],
    "gpsig_lr_kernel_diag": [_LR, _vp, _i64, _i32, _vp],
    "gpsig_seq_gram_levels_grad": [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, C.POINTER(C.c_double)],
    "gpsig_seq_gram_levels_stash": [_vp, _vp, _i64, _i64, _i32, _i32, _vp, C.POINTER(C.c_int64)],
    "gpsig_seq_gram_levels_grad_stash": [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32)],
    "gpsig_seq_diag_levels_grad": [_vp, _i64, _i32, _vp, _vp, C.POINTER(C.c_double)],
    "gpsig_tens_gram_levels_grad": [_vp, _i64, _i32, _vp, _vp, C.POINTER(C.c_double)],
    "gpsig_tens_vs_seq_levels_grad": [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, C.POINTER(C.c_double)],
}
_PLAIN = {
    "gpsig_abi_version": ([], C.c_int),
    "gpsig_lr_state_destroy": ([_vp], None),
    "gpsig_lr_state_sizes": ([_vp, _vp, C.POINTER(_i32), C.POINTER(_i32)], C.c_int),
    "gpsig_lr_state_export": ([_vp, _vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                               C.POINTER(SketchC)], C.c_int),
    "gpsig_tens_vs_seq_aux_elems": ([_P, _i64, _i64], _i64),
    "gpsig_seq_features_ld": ([_P, _i32], _i64),
    "gpsig_ctx_create": ([C.c_int, _vp, C.POINTER(_vp)], C.c_int),
    "gpsig_ctx_destroy": ([_vp], None),
    "gpsig_last_error": ([_vp], C.c_char_p),
    "gpsig_set_pointer_mode": ([_vp, C.c_int], C.c_int),
    "gpsig_sync": ([_vp], C.c_int),
    "gpsig_set_shard": ([_vp, C.c_int, C.c_int], C.c_int),
    "gpsig_set_option": ([_vp, C.c_char_p, C.c_int], C.c_int),
    "gpsig_symmetrize_owned_rows": ([_vp, _i32, _vp, _i64, _vp], C.c_int),
    "gpsig_symmetrize_compact_rows": ([_vp, _i32, _vp, _i64, _vp], C.c_int),
    "gpsig_timing_reset": ([_vp], C.c_int),
    "gpsig_timing_get": ([_vp, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(_i64)], C.c_int),
    "gpsig_timing_info": ([_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double)], C.c_int),
    "gpsig_clock_probe_start": ([_vp, C.c_double, _i32], C.c_int),
    "gpsig_clock_probe_read": ([_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)], C.c_int),
    "gpsig_clock_probe_xcds": ([_vp, C.POINTER(C.c_double), C.POINTER(_i32), _i32, C.POINTER(_i32)], C.c_int),
    "gpsig_graph_begin": ([_vp], C.c_int),
    "gpsig_graph_end": ([_vp, C.POINTER(_vp)], C.c_int),
    "gpsig_graph_launch": ([_vp, _vp], C.c_int),
    "gpsig_graph_destroy": ([_vp], None),
}
ALL_SYMBOLS = sorted(list(_KERNEL_FUNCS) + list(_PLAIN))

_lib = None


def load():
    """Load the HIP library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C gpsig_amd/csrc).  gpsig_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (args, res) in _PLAIN.items():
            f = getattr(lib, name)
            f.argtypes, f.restype = args, res
        for name, args in _KERNEL_FUNCS.items():
            f = getattr(lib, name)
            f.argtypes, f.restype = [_vp, _P] + args, C.c_int
        if lib.gpsig_abi_version() != 1:
            raise RuntimeError("libgpsig_hip.so ABI version mismatch")
        _lib = lib
    return _lib


class GpsigError(RuntimeError):
    pass


def raise_for(rc, msg):
    if rc == GPSIG_OK:
        return
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    raise GpsigError(msg)


class Context:
    """One gpsig_ctx: a (device, stream) pair plus its scratch memory."""

    def __init__(self, device=0, stream=0):
        lib = load()
        h = _vp()
        rc = lib.gpsig_ctx_create(int(device), _vp(stream) if stream else None, C.byref(h))
        if rc != GPSIG_OK:
            raise_for(rc, (lib.gpsig_last_error(None) or b"gpsig_ctx_create failed").decode())
        self._h, self._lib, self.device, self.stream = h, lib, int(device), int(stream or 0)
        # GPSIG_OPTIONS="name=value,name=value": gpsig_set_option calls for every context of the process (A/B runs of one build:
        # tools/gpu_round4.sh); an unknown name fails loudly
        for item in filter(None, (os.environ.get("GPSIG_OPTIONS") or "").split(",")):
            name, _, value = item.partition("=")
            self.check(lib.gpsig_set_option(h, name.strip().encode(), int(value)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpsig_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != GPSIG_OK:
            raise_for(rc, (self._lib.gpsig_last_error(self._h) or b"").decode())

    def call(self, name, params, *args):
        self.check(getattr(self._lib, name)(self._h, C.byref(params), *args))

    def set_pointer_mode(self, mode):
        self.check(self._lib.gpsig_set_pointer_mode(self._h, mode))

    def set_shard(self, index, count):
        self.check(self._lib.gpsig_set_shard(self._h, int(index), int(count)))

    def set_option(self, name, value):
        self.check(self._lib.gpsig_set_option(self._h, name.encode(), int(value)))

    def sync(self):
        self.check(self._lib.gpsig_sync(self._h))

    def symmetrize_compact_rows(self, dtype_id, half_ptr, n, out_ptr):
        self.check(self._lib.gpsig_symmetrize_compact_rows(self._h, int(dtype_id), half_ptr, int(n), out_ptr))

    def graph(self):
        """HIP-graph capture of the calls made in the with-block (include/gpsig_hip.h: gpsig_graph_begin):

            with ctx.graph() as g:
                ctx.call("gpsig_kernel_K", params, ...)      # recorded, not executed
            g.launch()                                        # replays them with one launch
        """
        return Graph(self)

    def timing_reset(self):
        self.check(self._lib.gpsig_timing_reset(self._h))

    def timing_get(self):
        ms, n, pairs = C.c_double(), _i64(), _i64()
        self.check(self._lib.gpsig_timing_get(self._h, C.byref(ms), C.byref(n), C.byref(pairs)))
        return ms.value, n.value, pairs.value


    def timing_info(self):
        """(kernel name or None, matrix-core flops) of the timed launches since the last reset."""
        k, f = C.c_char_p(), C.c_double()
        self.check(self._lib.gpsig_timing_info(self._h, C.byref(k), C.byref(f)))
        return (k.value.decode() if k.value else None), f.value

    def clock_probe_start(self, duration_ms, samples=64):
        """Sample the shader clock for `duration_ms` from now on, concurrently with whatever is launched next."""
        self.check(self._lib.gpsig_clock_probe_start(self._h, float(duration_ms), int(samples)))

    def clock_probe_read(self):
        """(mean, min, max) GHz between consecutive readings and the milliseconds they span."""
        a, b, c_, d = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        self.check(self._lib.gpsig_clock_probe_read(self._h, C.byref(a), C.byref(b), C.byref(c_), C.byref(d)))
        return a.value, b.value, c_.value, d.value

    def clock_probe_xcds(self):
        """[(XCD id, mean GHz)] of the last read: one sampling wavefront per XCD."""
        g, x, n = (C.c_double * 8)(), (_i32 * 8)(), _i32()
        self.check(self._lib.gpsig_clock_probe_xcds(self._h, g, x, 8, C.byref(n)))
        return [(int(x[k]), float(g[k])) for k in range(n.value)]


class Graph:
    def __init__(self, ctx):
        self.ctx, self._g = ctx, None

    def __enter__(self):
        self.ctx.check(self.ctx._lib.gpsig_graph_begin(self.ctx._h))
        return self

    def __exit__(self, exc_type, exc, tb):
        g = _vp()
        rc = self.ctx._lib.gpsig_graph_end(self.ctx._h, C.byref(g))
        if exc_type is None:
            self.ctx.check(rc)
            self._g = g
        return False

    def launch(self):
        if self._g is None:
            raise ValueError("the graph was not recorded")
        self.ctx.check(self.ctx._lib.gpsig_graph_launch(self.ctx._h, self._g))

    def destroy(self):
        """Release the executable graph (no replay may be in flight)."""
        g, self._g = getattr(self, "_g", None), None
        if g is not None:
            self.ctx._lib.gpsig_graph_destroy(g)

    def __del__(self):
        try:
            self.destroy()
        except Exception:        # interpreter shutdown: the library may already be gone
            pass


_holders = {}


def hold(device, stream):
    """A recorded call starts using the context of (device, stream): torch hands out side streams from a pool, so several
    recordings can end up on one stream and share its context."""
    key = (int(device), int(stream or 0))
    _holders[key] = _holders.get(key, 0) + 1


def release(device, stream):
    """The counterpart of hold(): the last holder drops the cached context and frees its scratch memory."""
    key = (int(device), int(stream or 0))
    n = _holders.get(key, 0) - 1
    if n > 0:
        _holders[key] = n
        return
    _holders.pop(key, None)
    ctx = _contexts.pop(key, None)
    if ctx is not None:
        ctx.close()


_contexts = {}


def context(device=0, stream=0):
    key = (int(device), int(stream or 0))
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = _contexts[key] = Context(*key)
    return ctx
