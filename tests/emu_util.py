"""Driver for the CPU lock-step emulator of the seq-gram kernel (tests/emu/emu_seq.cpp).

Test infrastructure only.  It builds the records the GPU prep kernel builds (restated in NumPy),
asks the shared C++ planning code for tasks / configs / geometry, runs the emulator, and returns
level tensors or finished kernel matrices, so the kernel's dataflow can be checked against the
oracle on a machine without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "gpsig_amd", "csrc")

BASE_IDS = {"linear": 0, "rbf": 1, "cosine": 2, "poly": 3, "mix": 4, "matern12": 5, "matern32": 6, "matern52": 7}
PRED_ALL, PRED_CIRCULANT, PRED_DIAG = 0, 1, 2


class SeqTask(C.Structure):
    _fields_ = [("y0", C.c_int32), ("x0", C.c_int32), ("nx", C.c_int32)]


class SeqGramArgs(C.Structure):
    _fields_ = [
        ("xrec", C.c_void_p), ("yrec", C.c_void_p), ("tasks", C.c_void_p),
        ("N1", C.c_int64), ("N2", C.c_int64), ("xrec_stride", C.c_int64), ("yrec_stride", C.c_int64),
        ("R1", C.c_int32), ("R2", C.c_int32), ("RS", C.c_int32), ("M", C.c_int32), ("order", C.c_int32), ("nslot", C.c_int32),
        ("issue_at", C.c_int32), ("slot_elems", C.c_int32), ("kind", C.c_int32),
        ("p0", C.c_double), ("p1", C.c_double),
        ("out", C.c_void_p), ("si", C.c_int64), ("sj", C.c_int64), ("sm", C.c_int64),
        ("ax", C.c_void_p), ("by", C.c_void_p), ("jitter_diag", C.c_double),
        ("sum_levels", C.c_int32), ("pred", C.c_int32), ("mirror", C.c_int32), ("use_glds", C.c_int32),
        ("compact", C.c_int32), ("spec", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(EMU_DIR, "libemu_seq.so")
        srcs = [os.path.join(EMU_DIR, "emu_seq.cpp")] + [os.path.join(CSRC, h) for h in
                                                        ("seq_core.hpp", "seq_args.hpp", "seq_configs.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + CSRC, "-o", so, srcs[0]])
        _lib = C.CDLL(so)
        _lib.emu_seq_gram.argtypes = [C.c_int] * 6 + [C.POINTER(SeqGramArgs), C.c_int]
        _lib.emu_seq_gram_ho.argtypes = [C.c_int] * 6 + [C.POINTER(SeqGramArgs), C.c_int]
        _lib.emu_build_tasks.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int64, C.c_int64, C.POINTER(SeqTask), C.c_int]
    return _lib


def select(Ry, d, M, allow_exact=True):
    cfg = (C.c_int * 5)()
    if lib().emu_select(Ry, d, M, int(allow_exact), cfg) < 0:
        raise NotImplementedError(f"no emulator config for Ry={Ry} d={d} M={M}")
    return dict(G=cfg[0], C=cfg[1], D=cfg[2], MMAX=cfg[3], exact=cfg[4])


def geometry(base, difference, L, D):
    out = (C.c_int * 4)()
    lib().emu_geometry(BASE_IDS[base], int(difference), L, D, 8, out)
    return dict(mode=out[0], rows=out[1], RS=out[2], rec_elems=out[3])


def build_records(Xs, geom, difference, D):
    """What the GPU prep kernel writes for already scaled sequences Xs (N, L, d)."""
    N, L, d = Xs.shape
    rec = np.zeros((N, geom["rec_elems"]), dtype=np.float64)
    rows = np.zeros((N, geom["rows"], geom["RS"]), dtype=np.float64)
    if geom["mode"] == 0:  # MODE_INC
        body = np.diff(Xs, axis=1) if difference else Xs
        rows[:, 1:1 + body.shape[1], :d] = body
    elif geom["mode"] == 1:  # MODE_PT_DIFF
        rows[:, :, :d] = Xs
    else:
        rows[:, 1:, :d] = Xs
    rec[:, :geom["rows"] * geom["RS"]] = rows.reshape(N, -1)
    return rec


def tasks_for(N1, N2, ypb, pred, max_run=64, shard=(0, 1), yrange=(0, -1)):
    n = lib().emu_build_tasks(N1, N2, ypb, pred, max_run, shard[0], shard[1], yrange[0], yrange[1], None, 0)
    arr = (SeqTask * max(n, 1))()
    lib().emu_build_tasks(N1, N2, ypb, pred, max_run, shard[0], shard[1], yrange[0], yrange[1], arr, n)
    return arr, n


def run(cfg, geom_x, geom_y, xrec, yrec, N1, N2, M, kind, p0, p1, out, si, sj, sm, ax, by, jitter_diag, sum_levels,
        pred, mirror, max_run=64, shard=(0, 1), yrange=(0, -1), out_offset=0, order=1, compact=0):
    tasks, nt = tasks_for(N1, N2, 64 // cfg["G"], pred, max_run, shard, yrange)
    A = SeqGramArgs()
    A.xrec, A.yrec, A.tasks = xrec.ctypes.data, yrec.ctypes.data, C.addressof(tasks)
    A.N1, A.N2 = N1, N2
    A.xrec_stride, A.yrec_stride = geom_x["rec_elems"], geom_y["rec_elems"]
    A.R1, A.R2, A.RS, A.M, A.order = geom_x["rows"], geom_y["rows"], geom_x["RS"], M, order
    A.nslot = lib().emu_ring_depth(cfg["G"], geom_x["rows"])
    A.issue_at = lib().emu_ring_issue_at(cfg["G"], geom_x["rows"])
    A.slot_elems = geom_x["rec_elems"]
    A.kind, A.p0, A.p1 = kind, p0, p1
    A.out, A.si, A.sj, A.sm = out.ctypes.data + 8 * out_offset, si, sj, sm
    A.ax = ax.ctypes.data if ax is not None else None
    A.by = by.ctypes.data if by is not None else None
    A.jitter_diag, A.sum_levels, A.pred, A.mirror, A.use_glds = jitter_diag, int(sum_levels), pred, int(mirror), 0
    A.compact = int(compact)
    if cfg.get("OMAX"):
        rc = lib().emu_seq_gram_ho(cfg["G"], cfg["C"], cfg["D"], cfg["MMAX"], cfg["OMAX"], geom_x["mode"], C.byref(A), nt)
    else:
        rc = lib().emu_seq_gram(cfg["G"], cfg["C"], cfg["D"], cfg["MMAX"], geom_x["mode"], cfg["exact"], C.byref(A), nt)
    if rc != 0:
        raise RuntimeError("emulator has no such config")
    return nt


def seq_levels(X1s, X2s, base, M, difference=True, base_params=(0.0, 0.0), diag_only=False, allow_exact=True,
               max_run=64, shard=(0, 1), fill=np.nan):
    """Unnormalised levels (M+1, N1, N2) [or (M+1, N) with diag_only] of scaled sequences through the
    emulator: the counterpart of gpsig_seq_gram_levels / gpsig_seq_diag_levels."""
    sym = X2s is None
    Y = X1s if sym else X2s
    N1, L1, d = X1s.shape
    N2, L2, _ = Y.shape
    gy = geometry(base, difference, L2, 4)
    cfg = select(gy["rows"], d, M, allow_exact)
    gx = geometry(base, difference, L1, cfg["D"])
    gy = geometry(base, difference, L2, cfg["D"])
    xrec = build_records(X1s, gx, difference, cfg["D"])
    yrec = xrec if sym else build_records(Y, gy, difference, cfg["D"])
    kind = BASE_IDS[base]
    if diag_only:
        out = np.full((M + 1, N1), fill)
        run(cfg, gx, gy, xrec, yrec, N1, N2, M, kind, *base_params, out, 1, 0, N1, None, None, 0.0, False, PRED_DIAG,
            False, max_run, shard)
    else:
        out = np.full((M + 1, N1, N2), fill)
        run(cfg, gx, gy, xrec, yrec, N1, N2, M, kind, *base_params, out, N2, 1, N1 * N2, None, None, 0.0, False,
            PRED_CIRCULANT if sym else PRED_ALL, sym, max_run, shard)
    return out, cfg


def kernel_K(X1s, X2s, base, M, variances, sigma, normalization, difference=True, base_params=(0.0, 0.0),
             return_levels=False, jitter=1e-6, allow_exact=True):
    """SignatureKernel.K on already scaled sequences via the emulator, with the on-chip epilogue:
    diag pass -> per-sequence factors -> main pass that normalises, weights and sums levels."""
    sym = X2s is None
    Y = X1s if sym else X2s
    N1, N2 = X1s.shape[0], Y.shape[0]
    w = sigma * np.asarray(variances, dtype=np.float64)
    if normalization:
        d1, cfg = seq_levels(X1s, None, base, M, difference, base_params, diag_only=True, allow_exact=allow_exact)
        d2 = d1 if sym else seq_levels(Y, None, base, M, difference, base_params, diag_only=True, allow_exact=allow_exact)[0]
        ax = np.ascontiguousarray((w[:, None] / np.sqrt(d1 + jitter)).T)
        by = np.ascontiguousarray((1.0 / np.sqrt(d2 + jitter)).T)
    else:
        ax = np.ascontiguousarray(np.tile(w[None, :], (N1, 1)))
        by = None
    L1, d = X1s.shape[1:]
    L2 = Y.shape[1]
    gy = geometry(base, difference, L2, 4)
    cfg = select(gy["rows"], d, M, allow_exact)
    gx = geometry(base, difference, L1, cfg["D"])
    gy = geometry(base, difference, L2, cfg["D"])
    xrec = build_records(X1s, gx, difference, cfg["D"])
    yrec = xrec if sym else build_records(Y, gy, difference, cfg["D"])
    out = np.full((M + 1, N1, N2) if return_levels else (N1, N2), np.nan)
    run(cfg, gx, gy, xrec, yrec, N1, N2, M, BASE_IDS[base], *base_params, out, N2, 1, N1 * N2, ax, by,
        jitter if (sym and normalization) else 0.0, not return_levels, PRED_CIRCULANT if sym else PRED_ALL, sym)
    return out


def kernel_K_owned_rows(Xs, base, M, variances, sigma, normalization, row_begin, row_end, jitter=1e-6, compact=False):
    """Emulator counterpart of gpsig_kernel_K_symm_rows[_compact]: the owned entries of rows [row_begin, row_end) of the
    symmetric, normalised, weighted, level-summed Gram; everything else is left at zero.  compact: (rows, N//2+1) blocks."""
    N, L, d = Xs.shape
    w = sigma * np.asarray(variances, dtype=np.float64)
    if normalization:
        dl, _ = seq_levels(Xs, None, base, M, diag_only=True)
        ax = np.ascontiguousarray((w[:, None] / np.sqrt(dl + jitter)).T)
        by = np.ascontiguousarray((1.0 / np.sqrt(dl + jitter)).T)
    else:
        ax, by = np.ascontiguousarray(np.tile(w[None, :], (N, 1))), None
    gy = geometry(base, True, L, 4)
    cfg = select(gy["rows"], d, M)
    g = geometry(base, True, L, cfg["D"])
    rec = build_records(Xs, g, True, cfg["D"])
    sj = N // 2 + 1 if compact else N
    out = np.zeros((row_end - row_begin, sj))
    # as the C API does: x index = column (si = 1), y index = row (sj), y indices >= row_end are invalid (N2 = row_end)
    run(cfg, g, g, rec, rec, N, row_end, M, BASE_IDS[base], 0.0, 0.0, out, 1, sj, 0, ax, by,
        jitter if normalization else 0.0, True, PRED_CIRCULANT, False, yrange=(row_begin, row_end), out_offset=-row_begin * sj,
        compact=compact)
    return out


HO_TABLE = [dict(G=64, C=1, D=4, MMAX=6, OMAX=6), dict(G=64, C=2, D=4, MMAX=5, OMAX=3), dict(G=16, C=2, D=4, MMAX=4, OMAX=4)]


def seq_levels_ho(X1s, X2s, base, M, order, cfg, difference=True, base_params=(0.0, 0.0), diag_only=False):
    """Unnormalised levels through the emulated HIGHER-ORDER kernel (signature_algs.py:37-74) with config `cfg`."""
    sym = X2s is None
    Y = X1s if sym else X2s
    N1, L1, d = X1s.shape
    N2, L2, _ = Y.shape
    gx = geometry(base, difference, L1, cfg["D"])
    gy = geometry(base, difference, L2, cfg["D"])
    assert gy["rows"] <= cfg["G"] * cfg["C"] and d <= cfg["D"] and M <= cfg["MMAX"] and order <= cfg["OMAX"]
    xrec = build_records(X1s, gx, difference, cfg["D"])
    yrec = xrec if sym else build_records(Y, gy, difference, cfg["D"])
    kind = BASE_IDS[base]
    if diag_only:
        out = np.full((M + 1, N1), np.nan)
        run(cfg, gx, gy, xrec, yrec, N1, N2, M, kind, *base_params, out, 1, 0, N1, None, None, 0.0, False, PRED_DIAG, False, order=order)
    else:
        out = np.full((M + 1, N1, N2), np.nan)
        run(cfg, gx, gy, xrec, yrec, N1, N2, M, kind, *base_params, out, N2, 1, N1 * N2, None, None, 0.0, False,
            PRED_CIRCULANT if sym else PRED_ALL, sym, order=order)
    return out
