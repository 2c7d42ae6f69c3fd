#!/usr/bin/env python3
"""Regenerates tests/golden/lowrank.{json,npz}: fixtures of LOW-RANK mode (gpsig/low_rank_calculations.py, signature_algs.py:162-222,
kernels.py:236-311).  The reference draws its landmarks and projections from TensorFlow's RNG inside the graph, so a fixture
carries the random objects themselves -- landmarks (scaled points), the jitter draw of low_rank_calculations.py:52, one sparse
projection per level >= 2 -- next to the seeded inputs and the outputs of the oracle's restatement given those objects.
Data only; produced by oracle/sigkern_oracle.py (LowRankOracle), like tests/golden/make_golden.py.

    python tests/golden/make_golden_lowrank.py
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import sigkern_oracle as O  # noqa: E402


def draw_sketch(rng, k1, k2, r, sparsity):
    """A projection of the reference's kinds (low_rank_calculations.py:104-193), stored by output column."""
    D = k1 * k2
    if sparsity == "lin":
        sel = rng.choice(D, size=r, replace=False)
        return dict(k1=k1, k2=k2, r=r, colptr=np.arange(r + 1), i1=sel % k1, i2=sel // k1, val=np.where(rng.random(r) <= 0.5, 1.0, -1.0))
    s = np.sqrt(float(D)) if sparsity == "sqrt" else float(D) / np.log(float(D))
    R = np.where(rng.random((D, r)) <= 1.0 / s, rng.standard_normal((D, r)), 0.0) * np.sqrt(s / r)
    colptr, i1, i2, val = [0], [], [], []
    for j in range(r):
        nz = np.nonzero(R[:, j])[0]
        i1.append(nz % k1); i2.append(nz // k1); val.append(R[nz, j]); colptr.append(colptr[-1] + nz.size)
    return dict(k1=k1, k2=k2, r=r, colptr=np.asarray(colptr), i1=np.concatenate(i1), i2=np.concatenate(i2), val=np.concatenate(val))


CASES, ARR = [], {}
rng = np.random.default_rng(2024)
for name, base, N, L, d, M, T, c, r, sp, norm, incr, lags in (
        ("lr_rbf_sqrt", "rbf", 9, 12, 3, 4, 5, 10, 8, "sqrt", True, False, 0),
        ("lr_rbf_log_incr", "rbf", 7, 9, 2, 3, 4, 8, 9, "log", False, True, 0),
        ("lr_rbf_lin_lags", "rbf", 6, 10, 2, 3, 3, 7, 5, "lin", True, False, 1),
        ("lr_matern32_sqrt", "matern32", 8, 8, 3, 2, 4, 9, 6, "sqrt", True, True, 0),
        ("lr_rbf_m1", "rbf", 5, 7, 2, 1, 3, 6, 4, "sqrt", False, False, 0)):
    kw = dict(input_dim=L * d, num_features=d, num_levels=M, base=base, normalization=norm, lengthscales=list(0.7 + rng.random(d)),
              variances=list(0.5 + rng.random(M + 1)))
    if lags:
        kw["num_lags"] = lags
    ko = O.SignatureKernelOracle(**{k: (np.asarray(v) if isinstance(v, list) else v) for k, v in kw.items()})
    de = d * (lags + 1)
    X = np.cumsum(0.3 * rng.standard_normal((N, L, d)), axis=1).reshape(N, -1)
    X2 = np.cumsum(0.3 * rng.standard_normal((4, L, d)), axis=1).reshape(4, -1)
    Z = rng.standard_normal((M * (M + 1) // 2, T, 2, de) if incr else (M * (M + 1) // 2, T, de))
    pts = ko._apply_scaling_and_lags_to_sequences(ko._seq3(X)).reshape(-1, de)                      # landmarks: scaled points of X
    S = pts[rng.choice(pts.shape[0], size=c, replace=False)]
    jd = 1e-6 * rng.random(c)
    sk, k2 = [], c
    for _ in range(2, M + 1):
        sk.append(draw_sketch(rng, c, k2, r, sp))
        k2 = r
    lo = O.LowRankOracle(ko, S, jd, [types.SimpleNamespace(**s_) for s_ in sk])
    outs = dict(K=lo.K(X), Kx=lo.K(X, X2, return_levels=True), Kzx=lo.K_tens_vs_seq(Z, X, increments=incr), Kzz=lo.K_tens(Z, increments=incr))
    if not norm:
        outs["Kdiag"] = lo.Kdiag(X)
    CASES.append(dict(name=name, kern=kw, increments=incr, sparsity=sp, num_components=c, rank_bound=r, outputs=sorted(outs)))
    for k_, v in dict(X=X, X2=X2, Z=Z, landmarks=S, jitter_diag=jd).items():
        ARR[f"{name}/{k_}"] = v
    for i, s_ in enumerate(sk):
        for k_ in ("colptr", "i1", "i2", "val"):
            ARR[f"{name}/sk{i}/{k_}"] = np.asarray(s_[k_])
        ARR[f"{name}/sk{i}/shape"] = np.asarray([s_["k1"], s_["k2"], s_["r"]])
    for k_, v in outs.items():
        ARR[f"{name}/out/{k_}"] = v
json.dump(CASES, open(os.path.join(HERE, "lowrank.json"), "w"), indent=1)
np.savez_compressed(os.path.join(HERE, "lowrank.npz"), **ARR)
print(len(CASES), "cases,", sum(v.nbytes for v in ARR.values()), "bytes")
