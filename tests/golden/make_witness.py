"""Independent witness values for the NON-LINEAR base kernels (round 6; VERDICT r5 "parity hardening").

The oracle (oracle/sigkern_oracle.py) is a transcription of the reference's graph; until now the only values it was pinned to that
were not its own were the signature identities of the linear kernel at order = num_levels.  This script computes, WITHOUT importing
anything from ``oracle`` or ``gpsig_amd``, in 50-digit arithmetic (mpmath), from the mathematical definitions alone:

* kappa(x, y) of rbf / matern12 / matern32 / matern52 / poly / mix / cosine from their closed forms on the scaled points
  (gpsig/kernels.py:820-993 say which closed form each class uses; lengthscales, lags and gamma as kernels.py:343-398 and lags.py:7-63
  define them: x / lengthscale, lagged copies by linear interpolation at max(t - lag, 0) on the grid t_i = i / (L - 1), times gamma);
* the level values of the order-1 algorithm as what signature_algs.py:8-35 MEANS: the sum over strictly increasing index tuples
  a_1 < .. < a_m, b_1 < .. < b_m of prod_l dM[a_l, b_l], dM the double increment of kappa along both sequences (literal loops over
  itertools.combinations -- no cumulative sums, no recursion);
* the tensor-vs-sequence levels (signature_algs.py:101-127) as the sum over increasing time tuples t_1 < .. < t_i of
  prod_j d kappa(z_{k0+j}, x)[t_j], with increments the difference of the component's two points (kernels.py:329-330);
* the tensor-vs-tensor levels (signature_algs.py:76-99) as products of component values;
* normalisation of the symmetric Gram as kernels.py:430-433 defines it (jitter on the diagonal, then the cosine normalisation).

Outputs rounded to float64 go to tests/golden/witness.npz + witness.json; tests/test_oracle.py holds the oracle to them (1e-10) and
tests/test_gpu_parity.py the HIP path.  Run from the repo root: python tests/golden/make_witness.py"""
import itertools
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))
JITTER = mp.mpf("1e-6")                                     # gpflow settings.jitter, kernels.py:431


def mpv(a):
    return [mp.mpf(float(v)) for v in a]


def dot(x, y):
    return mp.fsum(a * b for a, b in zip(x, y))


def sqdist(x, y):
    return mp.fsum((a - b) ** 2 for a, b in zip(x, y))


def kappa(name, x, y, par):
    if name == "rbf":
        return mp.exp(-sqdist(x, y) / 2)
    if name == "matern12":
        return mp.exp(-mp.sqrt(sqdist(x, y)))
    if name == "matern32":
        r = mp.sqrt(sqdist(x, y))
        return (1 + mp.sqrt(3) * r) * mp.exp(-mp.sqrt(3) * r)
    if name == "matern52":
        r = mp.sqrt(sqdist(x, y))
        return (1 + mp.sqrt(5) * r + mp.mpf(5) / 3 * r * r) * mp.exp(-mp.sqrt(5) * r)
    if name == "poly":
        return (dot(x, y) + par["gamma"]) ** par["degree"]
    if name == "mix":
        return par["mixing"] * mp.exp(-sqdist(x, y) / 2) + (1 - par["mixing"]) * dot(x, y)
    if name == "cosine":
        return dot(x, y) / (mp.sqrt(dot(x, x)) * mp.sqrt(dot(y, y)))
    raise KeyError(name)


def lagged_scaled(seq, lengthscales, lags, gamma):
    """One sequence (L, d) of floats -> L points of (num_lags + 1) * d mp numbers."""
    L, d = seq.shape
    x = [mpv(r) for r in seq]
    ls = mpv(lengthscales) if lengthscales is not None else [mp.mpf(1)] * d
    copies = [x]
    for lag in (lags if lags is not None else []):
        lag = mp.mpf(float(lag))
        cp = []
        for i in range(L):
            tq = max(mp.mpf(i) / (L - 1) - lag, mp.mpf(0))
            pos = tq * (L - 1)
            left = min(int(mp.floor(pos)), L - 2)
            w = pos - left
            cp.append([x[left][f] + w * (x[left + 1][f] - x[left][f]) for f in range(d)])
        copies.append(cp)
    g = mpv(gamma) if lags is not None else [mp.mpf(1)]
    out = []
    for i in range(L):
        row = []
        for c, cp in enumerate(copies):
            row += [g[c] * cp[i][f] / ls[f] for f in range(d)]
        out.append(row)
    return out


def scaled_tensor_point(z, lengthscales, nlag_copies, gamma):
    d = len(z) // nlag_copies
    ls = mpv(lengthscales) if lengthscales is not None else [mp.mpf(1)] * d
    g = mpv(gamma) if nlag_copies > 1 else [mp.mpf(1)]
    z = mpv(z)
    if lengthscales is None:                                 # kernels.py:374 / :391: tensors are scaled (and weighted by gamma) only with lengthscales
        return z
    return [g[c] * z[c * d + f] / ls[f] for c in range(nlag_copies) for f in range(d)]


def tuple_levels(dM, M):
    """sum over a_1<..<a_m, b_1<..<b_m of prod dM[a_l][b_l], m = 0..M."""
    l1, l2 = len(dM), len(dM[0])
    out = [mp.mpf(1)]
    for m in range(1, M + 1):
        tot = mp.mpf(0)
        for A in itertools.combinations(range(l1), m):
            for B in itertools.combinations(range(l2), m):
                p = mp.mpf(1)
                for a, b in zip(A, B):
                    p *= dM[a][b]
                tot += p
        out.append(tot)
    return out


def seq_levels(name, xs, ys, M, par):
    k = [[kappa(name, a, b, par) for b in ys] for a in xs]
    dM = [[k[a + 1][b + 1] + k[a][b] - k[a][b + 1] - k[a + 1][b] for b in range(len(ys) - 1)] for a in range(len(xs) - 1)]
    return tuple_levels(dM, M)


def chain_levels(name, zcomp, xs, M, par, increments):
    """zcomp: lt components, each a point (or a pair of points); K_i = sum_{t_1<..<t_i} prod_j dk[k0+j][t_j]."""
    def comp_vals(zc):
        if increments:
            return [kappa(name, zc[1], x, par) - kappa(name, zc[0], x, par) for x in xs]
        return [kappa(name, zc, x, par) for x in xs]
    vals = [comp_vals(zc) for zc in zcomp]
    dk = [[v[t + 1] - v[t] for t in range(len(xs) - 1)] for v in vals]
    out, k0 = [mp.mpf(1)], 0
    for i in range(1, M + 1):
        tot = mp.mpf(0)
        for Tt in itertools.combinations(range(len(xs) - 1), i):
            p = mp.mpf(1)
            for j, t in enumerate(Tt):
                p *= dk[k0 + j][t]
            tot += p
        out.append(tot)
        k0 += i
    return out


def tens_levels(name, za, zb, M, par, increments):
    def val(a, b):
        if increments:                                       # kernels.py:276-277: the 4-term difference
            return kappa(name, a[1], b[1], par) + kappa(name, a[0], b[0], par) - kappa(name, a[0], b[1], par) - kappa(name, a[1], b[0], par)
        return kappa(name, a, b, par)
    out, k0 = [mp.mpf(1)], 0
    for i in range(1, M + 1):
        p = mp.mpf(1)
        for j in range(i):
            p *= val(za[k0 + j], zb[k0 + j])
        out.append(p)
        k0 += i
    return out


CASES = [
    # name, base, d, L1, L2, M, lengthscales, lags, params
    ("rbf_plain", "rbf", 2, 5, 6, 3, None, None, {}),
    ("rbf_ls", "rbf", 3, 6, 5, 3, [0.8, 1.3, 2.1], None, {}),
    ("rbf_lags", "rbf", 2, 6, 6, 3, [1.1, 0.7], [0.13], {}),
    ("matern12_ls", "matern12", 2, 5, 5, 3, [0.9, 1.4], None, {}),
    ("matern32_ls", "matern32", 3, 5, 6, 3, [1.2, 0.8, 1.7], None, {}),
    ("matern52_plain", "matern52", 2, 6, 5, 3, None, None, {}),
    ("matern32_lags", "matern32", 2, 6, 5, 3, [1.3, 0.9], [0.27], {}),
    ("poly_ls", "poly", 2, 5, 5, 3, [1.5, 2.0], None, {"gamma": 0.7, "degree": 3.0}),
    ("mix_ls", "mix", 3, 5, 5, 3, [1.1, 1.6, 0.9], None, {"mixing": 0.35}),
    ("cosine_plain", "cosine", 3, 5, 6, 3, None, None, {}),
    ("rbf_m4", "rbf", 2, 6, 6, 4, [1.4, 1.0], None, {}),
]


def main():
    rng = np.random.default_rng(20261001)
    arrays, meta = {}, []
    for name, base, d, L1, L2, M, ls, lags, par in CASES:
        nx, ny, T = 3, 2, 3
        X = np.cumsum(0.6 * rng.standard_normal((nx, L1, d)), axis=1) + (1.0 if base == "cosine" else 0.0)
        Y = np.cumsum(0.6 * rng.standard_normal((ny, L2, d)), axis=1) + (1.0 if base == "cosine" else 0.0)
        nl = 1 + (len(lags) if lags is not None else 0)
        gamma = None
        if lags is not None:
            g = 1.0 / np.arange(1, nl + 1)
            gamma = (g / g.sum()) * np.array([1.0, 1.35][:nl])          # not the default values: the weights must matter
        lt = M * (M + 1) // 2
        Z = 0.8 * rng.standard_normal((lt, T, d * nl)) + (1.0 if base == "cosine" else 0.0)
        Zi = 0.8 * rng.standard_normal((lt, T, 2, d * nl)) + (1.0 if base == "cosine" else 0.0)
        mpar = {k: mp.mpf(float(v)) for k, v in par.items()}
        xs = [lagged_scaled(x, ls, lags, gamma) for x in X]
        ys = [lagged_scaled(y, ls, lags, gamma) for y in Y]
        # sequence vs sequence: cross Gram levels, symmetric Gram levels (off-diagonal and diagonal), normalised symmetric sum
        cross = [[seq_levels(base, a, b, M, mpar) for b in ys] for a in xs]
        sym = [[seq_levels(base, a, b, M, mpar) for b in xs] for a in xs]
        arrays[name + "/X"], arrays[name + "/Y"], arrays[name + "/Z"], arrays[name + "/Zi"] = X, Y, Z, Zi
        arrays[name + "/K_cross_levels"] = np.array([[[float(cross[i][j][m]) for j in range(ny)] for i in range(nx)] for m in range(M + 1)])
        arrays[name + "/K_symm_levels"] = np.array([[[float(sym[i][j][m]) for j in range(nx)] for i in range(nx)] for m in range(M + 1)])
        var = 0.5 + rng.random(M + 1)
        Kn = [[mp.mpf(0)] * nx for _ in range(nx)]
        for m in range(M + 1):
            for i in range(nx):
                for j in range(nx):
                    num = sym[i][j][m] + (JITTER if i == j else 0)
                    den = mp.sqrt(sym[i][i][m] + JITTER) * mp.sqrt(sym[j][j][m] + JITTER)
                    Kn[i][j] += mp.mpf(float(var[m])) * num / den
        arrays[name + "/variances"] = var
        arrays[name + "/K_symm_normalised"] = np.array([[float(Kn[i][j]) for j in range(nx)] for i in range(nx)])
        # tensors
        zs = [[scaled_tensor_point(Z[k, t], ls, nl, gamma) for k in range(lt)] for t in range(T)]
        zis = [[(scaled_tensor_point(Zi[k, t, 0], ls, nl, gamma), scaled_tensor_point(Zi[k, t, 1], ls, nl, gamma)) for k in range(lt)] for t in range(T)]
        for tag, zz, inc in (("", zs, False), ("_incr", zis, True)):
            kzx = [[chain_levels(base, zz[t], xs[n], M, mpar, inc) for n in range(nx)] for t in range(T)]
            kzz = [[tens_levels(base, zz[t], zz[u], M, mpar, inc) for u in range(T)] for t in range(T)]
            arrays[name + "/Kzx%s_levels" % tag] = np.array([[[float(kzx[t][n][m]) for n in range(nx)] for t in range(T)] for m in range(M + 1)])
            arrays[name + "/Kzz%s_levels" % tag] = np.array([[[float(kzz[t][u][m]) for u in range(T)] for t in range(T)] for m in range(M + 1)])
        meta.append({"name": name, "base": base, "d": d, "L1": L1, "L2": L2, "M": M, "lengthscales": ls, "lags": lags,
                     "gamma": None if gamma is None else [float(v) for v in gamma], "params": par})
        print(name, "done")
    np.savez_compressed(os.path.join(HERE, "witness.npz"), **arrays)
    with open(os.path.join(HERE, "witness.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
