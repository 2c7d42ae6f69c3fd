"""ctypes driver of tests/emu/emu_grad.cpp: the per-pair gradient code of gpsig_amd/csrc/grad_core.hpp run on the CPU.
Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "gpsig_amd", "csrc")
BASE_IDS = {"linear": 0, "rbf": 1, "cosine": 2, "poly": 3, "mix": 4, "matern12": 5, "matern32": 6, "matern52": 7}
_P = C.POINTER(C.c_double)
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(EMU_DIR, "libemu_grad.so")
        srcs = [os.path.join(EMU_DIR, "emu_grad.cpp"), os.path.join(CSRC, "grad_core.hpp"), os.path.join(CSRC, "grad_wave_core.hpp"),
                os.path.join(CSRC, "seq_core.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + CSRC, "-o", so, srcs[0]])
        _lib = C.CDLL(so)
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_P)


def lattice_mode(base, difference):
    return 2 if not difference else (0 if base == "linear" else 1)


def seq_grad(X, Y, G, M, base, difference, p0=0.0, p1=0.0, diag=False):
    """-> levels, gX, gY (None unless Y), g_p0"""
    X = np.ascontiguousarray(X, np.float64)
    Y = None if Y is None else np.ascontiguousarray(Y, np.float64)
    G = np.ascontiguousarray(G, np.float64)
    N1, L1, d = X.shape
    N2, L2 = (N1, L1) if Y is None else Y.shape[:2]
    gX, gY, lev, gb = np.zeros_like(X), (None if Y is None else np.zeros_like(Y)), np.zeros_like(G), np.zeros(2)
    lib().emu_seq_grad(_ptr(X), _ptr(Y), N1, N2, L1, L2, d, M, BASE_IDS[base], lattice_mode(base, difference), C.c_double(p0), C.c_double(p1),
                       int(diag), _ptr(G), _ptr(gX), _ptr(gY), _ptr(lev), _ptr(gb))
    return lev, gX, gY, gb[0]


def tvs_grad(Z, X, G, M, base, difference, increments, p0=0.0, p1=0.0, fused=False):
    Z, X, G = (np.ascontiguousarray(a, np.float64) for a in (Z, X, G))
    T, (N, L, d) = Z.shape[1], X.shape
    gZ, gX, lev, gb = np.zeros_like(Z), np.zeros_like(X), np.zeros_like(G), np.zeros(2)
    rc = lib().emu_tvs_grad(_ptr(Z), _ptr(X), T, N, L, d, M, BASE_IDS[base], int(increments), int(difference), C.c_double(p0), C.c_double(p1),
                            _ptr(G), _ptr(gZ), _ptr(gX), _ptr(lev), _ptr(gb), int(fused))
    if rc != 0:
        raise NotImplementedError("no such emulator variant")
    return lev, gZ, gX, gb[0]


def tens_grad(Z, G, M, base, increments, p0=0.0, p1=0.0, row_owned=False):
    Z, G = np.ascontiguousarray(Z, np.float64), np.ascontiguousarray(G, np.float64)
    T, d = Z.shape[1], Z.shape[-1]
    gZ, gb = np.zeros_like(Z), np.zeros(2)
    if lib().emu_tens_grad(_ptr(Z), T, d, M, BASE_IDS[base], int(increments), C.c_double(p0), C.c_double(p1), _ptr(G), _ptr(gZ), _ptr(gb), int(row_owned)):
        raise NotImplementedError("no such emulator variant")
    return gZ, gb[0]


def seq_grad_wave(X, Y, G, M, base, difference, p0=0.0, p1=0.0, diag=False, group=16, cols=4, scratch_free=False):
    """The wave formulation (skewed forward sweep, oppositely skewed backward sweep).  -> gX, gY, g_p0
    scratch_free: False = forward lattice kept, True = forward recursion undone with the gradient of the register side formed
    in the sweep, "lam" = forward recursion undone with Lam out and the per-pair contraction."""
    X = np.ascontiguousarray(X, np.float64)
    Y = None if Y is None else np.ascontiguousarray(Y, np.float64)
    G = np.ascontiguousarray(G, np.float64)
    N1, L1, d = X.shape
    N2, L2 = (N1, L1) if Y is None else Y.shape[:2]
    gX, gY, gb = np.zeros_like(X), (None if Y is None else np.zeros_like(Y)), np.zeros(2)
    args = (_ptr(X), _ptr(Y), N1, N2, L1, L2, d, M, BASE_IDS[base], lattice_mode(base, difference), C.c_double(p0), C.c_double(p1),
            int(diag), _ptr(G), _ptr(gX), _ptr(gY), _ptr(gb), int(group), int(cols))
    if scratch_free is True:
        rc = lib().emu_seq_grad_wave2(*args)
    else:
        rc = lib().emu_seq_grad_wave(*args, int(scratch_free == "lam"))
    if rc != 0:
        raise NotImplementedError("wave emulator: unsupported shape")
    return gX, gY, gb[0]
