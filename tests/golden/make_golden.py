#!/usr/bin/env python3
"""Regenerates tests/golden/cases.{json,npz}.

The reference (TensorFlow 1.15 / GPflow 1.5.1) cannot be imported in this image and ships no
golden vectors of its own, so these fixtures are produced by the NumPy
restatement in oracle/sigkern_oracle.py, AFTER that restatement has been pinned by the
notebook's signature identities (tests/test_oracle.py).  A fixture is data only: seeded inputs,
constructor arguments, and fp64 outputs.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import sigkern_oracle as O  # noqa: E402


def paths(rng, N, L, d, kind):
    if kind == "white":  # notebook cell 4
        return rng.standard_normal((N, L, d))
    return np.cumsum(0.3 * rng.standard_normal((N, L, d)), axis=1)  # random walks


CASES = []
ARR = {}


def add(name, method, kern_kw, X=None, X2=None, Z=None, call_kw=None, W=None):
    call_kw = dict(call_kw or {})
    kw = dict(kern_kw)
    k = O.SignatureKernelOracle(**kw)
    args = {}
    if X is not None:
        args["X"] = X.reshape(X.shape[0], -1)
    if X2 is not None:
        args["X2"] = X2.reshape(X2.shape[0], -1)
    if Z is not None:
        args["Z"] = Z
    if W is not None:
        args["W"] = W
    if method == "K":
        out = k.K(args["X"], args.get("X2"), **call_kw)
    elif method == "Kdiag":
        out = k.Kdiag(args["X"], **call_kw)
    elif method == "K_tens":
        out = k.K_tens(Z, **call_kw)
    elif method == "K_tens_vs_seq":
        out = k.K_tens_vs_seq(Z, args["X"], **call_kw)
    elif method == "K_tens_n_seq_covs":
        out = k.K_tens_n_seq_covs(Z, args["X"], **call_kw)
    elif method == "K_seq_n_seq_covs":
        out = k.K_seq_n_seq_covs(args["X"], args["X2"], **call_kw)
    elif method == "Kuu_Kuf_Kff_tensors":
        out = O.inducing_tensors_Kuu_Kuf_Kff(k, Z, args["X"], W=W, **call_kw)
    elif method == "Kuu_Kuf_Kff_sequences":
        out = O.inducing_sequences_Kuu_Kuf_Kff(k, args["X"], args["X2"], W=W, **call_kw)
    else:
        raise ValueError(method)
    outs = list(out) if isinstance(out, tuple) else [out]
    for key, val in args.items():
        ARR[f"{name}/{key}"] = np.asarray(val, dtype=np.float64)
    for i, o in enumerate(outs):
        ARR[f"{name}/out{i}"] = np.asarray(o, dtype=np.float64)
    kern_json = {k_: (v.tolist() if isinstance(v, np.ndarray) else v) for k_, v in kern_kw.items()}
    CASES.append({"name": name, "method": method, "kern": kern_json, "call": call_kw, "n_out": len(outs),
                  "has": sorted(args.keys())})


def main():
    rng = np.random.default_rng(20260929)

    # (1) notebook shapes (cells 4, 11, 15, 21, 27) at reduced N/T: linear, order=M, unnormalised
    L, d, M = 50, 3, 5
    X = paths(rng, 40, L, d, "white")
    Z = rng.standard_normal((M * (M + 1) // 2, 30, d))
    nb = dict(input_dim=L * d, num_features=d, num_levels=M, base="linear", order=M, normalization=False)
    add("nb_K", "K", nb, X=X)
    add("nb_Kzx", "K_tens_vs_seq", nb, X=X, Z=Z)
    add("nb_Kzz", "K_tens", nb, Z=Z)

    # (2) BASELINE config 1: N=64 L=32 d=3 M=4, order in {1, 4}, normalisation on/off
    L, d, M = 32, 3, 4
    X = paths(rng, 64, L, d, "white")
    for order in (1, 4):
        for norm in (False, True):
            kw = dict(input_dim=L * d, num_features=d, num_levels=M, base="linear", order=order, normalization=norm)
            add(f"c1_o{order}_n{int(norm)}", "K", kw, X=X)
    add("c1_levels", "K", dict(input_dim=L * d, num_features=d, num_levels=M, base="linear", order=1,
                               normalization=True), X=X[:24], call_kw=dict(return_levels=True))

    # (3) every base kernel, X2 != None, ragged lengths (L1 != L2), lengthscales/variances, levels
    d, M = 4, 4
    L1, L2 = 21, 13
    Xa = paths(rng, 19, L1, d, "walk")
    Xb = paths(rng, 11, L2, d, "walk")
    ls = 0.5 + rng.random(d)
    var = 0.5 + rng.random(M + 1)
    bases = [("linear", {}), ("rbf", {}), ("cosine", {}), ("poly", dict(gamma=0.7, degree=3.0)),
             ("mix", dict(mixing=0.3)), ("matern12", {}), ("matern32", {}), ("matern52", {})]
    for base, bp in bases:
        for norm in (False, True):
            kw = dict(input_dim=L1 * d, num_features=d, num_levels=M, base=base, base_params=bp, order=1,
                      normalization=norm, lengthscales=ls, variances=var)
            add(f"base_{base}_n{int(norm)}_sym", "K", kw, X=Xa)
            add(f"base_{base}_n{int(norm)}_x2", "K", kw, X=Xa, X2=Xb)
    kw = dict(input_dim=L1 * d, num_features=d, num_levels=M, base="rbf", order=1, normalization=True,
              lengthscales=ls, variances=var)
    add("rbf_x2_levels", "K", kw, X=Xa, X2=Xb, call_kw=dict(return_levels=True))
    kwn = dict(kw, normalization=False)
    add("rbf_Kdiag_levels", "Kdiag", kwn, X=Xa, call_kw=dict(return_levels=True))
    add("rbf_Kdiag", "Kdiag", kwn, X=Xa)
    add("rbf_Kdiag_norm", "Kdiag", kw, X=Xa)
    add("lin_nodiff", "K", dict(input_dim=L1 * d, num_features=d, num_levels=3, base="linear", order=1,
                                normalization=True, difference=False), X=0.3 * Xa)
    add("rbf_nodiff_x2", "K", dict(input_dim=L1 * d, num_features=d, num_levels=3, base="rbf", order=1,
                                   normalization=False, difference=False), X=Xa, X2=Xb)
    add("rbf_no_lengthscales", "K", dict(input_dim=L1 * d, num_features=d, num_levels=M, base="rbf", order=1,
                                         normalization=True, lengthscales=None), X=Xa)

    # (4) higher order 1 < D <= M, sequences and tensors
    for order in (2, 3):
        kw = dict(input_dim=L1 * d, num_features=d, num_levels=M, base="rbf", order=order, normalization=True,
                  lengthscales=ls, variances=var)
        add(f"rbf_order{order}_sym", "K", kw, X=Xa)
        add(f"rbf_order{order}_x2", "K", kw, X=Xa, X2=Xb)

    # (5) inducing tensors: Kzz, Kzx, with and without increments, all three covs, learn_weights mixing
    T = 9
    lt = M * (M + 1) // 2
    Zp = rng.standard_normal((lt, T, d))
    Zi = rng.standard_normal((lt, T, 2, d))
    for base in ("linear", "rbf", "matern32"):
        for order in (1, 2):
            kw = dict(input_dim=L1 * d, num_features=d, num_levels=M, base=base, order=order, normalization=True,
                      lengthscales=ls, variances=var)
            add(f"tens_{base}_o{order}_Kzz", "K_tens", kw, Z=Zp)
            add(f"tens_{base}_o{order}_Kzz_incr", "K_tens", kw, Z=Zi, call_kw=dict(increments=True))
            add(f"tens_{base}_o{order}_Kzx", "K_tens_vs_seq", kw, X=Xa, Z=Zp)
            add(f"tens_{base}_o{order}_Kzx_incr_lv", "K_tens_vs_seq", kw, X=Xa, Z=Zi,
                call_kw=dict(increments=True, return_levels=True))
            add(f"tens_{base}_o{order}_covs", "K_tens_n_seq_covs", kw, X=Xa, Z=Zi, call_kw=dict(increments=True))
    kw = dict(input_dim=L1 * d, num_features=d, num_levels=M, base="rbf", order=1, normalization=False,
              lengthscales=ls, variances=var)
    add("tens_rbf_covs_full_nonorm", "K_tens_n_seq_covs", kw, X=Xa, Z=Zp, call_kw=dict(full_X_cov=True))
    kw["normalization"] = True
    add("tens_rbf_covs_full_norm_lv", "K_tens_n_seq_covs", kw, X=Xa, Z=Zp,
        call_kw=dict(full_X_cov=True, return_levels=True))
    W = np.tile(np.eye(T)[None], [M, 1, 1]) + 0.1 * rng.standard_normal((M, T, T))
    add("feat_tensors_W", "Kuu_Kuf_Kff_tensors", kw, X=Xa, Z=Zi, W=W, call_kw=dict(increments=True, jitter=1e-6))
    add("feat_tensors", "Kuu_Kuf_Kff_tensors", kw, X=Xa, Z=Zp, call_kw=dict(jitter=1e-6))

    # (6) inducing sequences
    Zs = paths(rng, 7, 10, d, "walk")
    kwq = dict(input_dim=L1 * d, num_features=d, num_levels=M, base="rbf", order=1, normalization=True,
               lengthscales=ls, variances=var)
    add("seqcovs_norm", "K_seq_n_seq_covs", kwq, X=Zs, X2=Xa)
    add("seqcovs_nonorm_full", "K_seq_n_seq_covs", dict(kwq, normalization=False), X=Zs, X2=Xa,
        call_kw=dict(full_X2_cov=True))
    Ws = np.tile(np.eye(7)[None], [M, 1, 1]) + 0.1 * rng.standard_normal((M, 7, 7))
    add("feat_sequences_W", "Kuu_Kuf_Kff_sequences", kwq, X=Zs, X2=Xa, W=Ws, call_kw=dict(jitter=1e-6))

    # (7) lags
    kwl = dict(input_dim=L1 * d, num_features=d, num_levels=3, base="rbf", order=1, normalization=True,
               lengthscales=ls, num_lags=2)
    add("lags_sym", "K", kwl, X=Xa)
    add("lags_x2", "K", kwl, X=Xa, X2=Xb)
    Zl = rng.standard_normal((6, T, d * 3))
    add("lags_Kzx", "K_tens_vs_seq", kwl, X=Xa, Z=Zl)

    # (8) edge shapes: N=1, L=2 (a single increment), d=1, M=1
    Xe = paths(rng, 1, 2, 1, "white")
    add("edge_min", "K", dict(input_dim=2, num_features=1, num_levels=1, base="linear", normalization=False), X=Xe)
    Xe = paths(rng, 5, 3, 2, "white")
    add("edge_short", "K", dict(input_dim=6, num_features=2, num_levels=6, base="rbf", normalization=True), X=Xe)
    # repeated-observation padding (preprocessing.py:22-24) => zero increments
    Xp = paths(rng, 6, 12, 3, "walk")
    Xp[:, 8:] = Xp[:, 7:8]
    add("edge_padded", "K", dict(input_dim=36, num_features=3, num_levels=4, base="linear", normalization=True), X=Xp)
    # longer than one 64-lane wave, and d = 16
    Xl = paths(rng, 6, 100, 2, "walk")
    add("edge_long", "K", dict(input_dim=200, num_features=2, num_levels=4, base="rbf", normalization=True), X=Xl)
    Xw = 0.5 * paths(rng, 6, 20, 16, "walk")
    add("edge_wide", "K", dict(input_dim=320, num_features=16, num_levels=4, base="linear", normalization=True), X=Xw)

    np.savez_compressed(os.path.join(HERE, "cases.npz"), **ARR)
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(CASES, f, indent=1)
    print(f"wrote {len(CASES)} cases, {sum(a.nbytes for a in ARR.values()) / 1e6:.2f} MB raw")


if __name__ == "__main__":
    main()
