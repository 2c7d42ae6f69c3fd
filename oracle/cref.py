"""ctypes access to oracle/sigkern_ref.c (TEST INFRASTRUCTURE: the C restatement of the reference's level recursions, built
with gcc -O3 -fopenmp into oracle/_build/; used by tests/test_oracle.py and by bench.py's cpu_baseline leg only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "sigkern_ref.c")
LIB = os.path.join(_HERE, "_build", "libsigkern_ref.so")
_lib = None


def build(force=False):
    """gcc -O3 -fopenmp oracle/sigkern_ref.c -> oracle/_build/libsigkern_ref.so (rebuilt when the source is newer)."""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"])
    return LIB


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        lib.sigkern_seq_levels.argtypes = [dp, dp] + [C.c_int] * 8 + [dp]
        lib.sigkern_tens_vs_seq_levels.argtypes = [dp, dp] + [C.c_int] * 8 + [dp]
        lib.sigkern_ref_threads.restype = C.c_int
        _lib = lib
    return _lib


BASE = {"linear": 0, "rbf": 1}


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def seq_levels(X, Y, num_levels, base="linear", difference=True):
    """SignatureKernel._K_seq on SCALED sequences X (n1, L1, d), Y (n2, L2, d) -> (M+1, n1, n2); first-order algorithm."""
    X, Y = np.ascontiguousarray(X, dtype=np.float64), np.ascontiguousarray(Y, dtype=np.float64)
    out = np.empty((num_levels + 1, X.shape[0], Y.shape[0]))
    load().sigkern_seq_levels(_p(X), _p(Y), X.shape[0], Y.shape[0], X.shape[1], Y.shape[1], X.shape[2], num_levels, BASE[base],
                              int(bool(difference)), _p(out))
    return out


def tens_vs_seq_levels(Z, X, num_levels, base="linear", difference=True):
    """SignatureKernel._K_tens_vs_seq on SCALED Z (lt, T, d) or (lt, T, 2, d) and X (n, L, d) -> (M+1, T, n)."""
    Z, X = np.ascontiguousarray(Z, dtype=np.float64), np.ascontiguousarray(X, dtype=np.float64)
    E = 2 if Z.ndim == 4 else 1
    out = np.empty((num_levels + 1, Z.shape[1], X.shape[0]))
    load().sigkern_tens_vs_seq_levels(_p(Z), _p(X), Z.shape[1], X.shape[0], X.shape[1], X.shape[2], num_levels, E, BASE[base],
                                      int(bool(difference)), _p(out))
    return out


def threads():
    return int(load().sigkern_ref_threads())
