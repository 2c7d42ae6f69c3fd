"""Randomness of the low-rank signature-kernel algorithms (reference: gpsig/low_rank_calculations.py).

The reference draws its landmarks and random projections with TensorFlow's RNG inside the graph; that stream
cannot be reproduced, so parity with it is statistical only.  Here the random objects are built on the host with
NumPy's Generator and handed to the HIP kernels as plain arrays:

* landmarks: `num_components` rows drawn without replacement (low_rank_calculations.py:12-20, :47-48);
* one sketch per signature level >= 2, shared by every row it is applied to (as the reference's single R matrix
  per call): a sparse map from the k1*k2 coordinates of a row-wise Kronecker product to `rank_bound` outputs,
  stored by output column: column j sums val * A[i1] * B[i2] over its entries.
    - 'lin'  (low_rank_calculations.py:104-127): rank_bound coordinate pairs drawn without replacement, Rademacher signs;
    - 'sqrt' / 'log' (:152-193): very sparse Johnson-Lindenstrauss matrix with N(0,1) entries kept with probability
      1/s, s = sqrt(D) or D/log(D), scaled by sqrt(s/rank_bound).
"""
import numpy as np


class Sketch:
    """colptr (r+1,), i1, i2 (nnz,) int32, val (nnz,) float64; k1, k2 input widths; r outputs."""

    def __init__(self, k1, k2, r, colptr, i1, i2, val):
        self.k1, self.k2, self.r = int(k1), int(k2), int(r)
        self.colptr = np.ascontiguousarray(colptr, dtype=np.int32)
        self.i1 = np.ascontiguousarray(i1, dtype=np.int32)
        self.i2 = np.ascontiguousarray(i2, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float64)

    def apply(self, A, B):
        """NumPy statement of what the HIP kernel computes: (..., k1), (..., k2) -> (..., r)."""
        out = np.zeros(A.shape[:-1] + (self.r,), dtype=np.result_type(A, B))
        for j in range(self.r):
            sl = slice(self.colptr[j], self.colptr[j + 1])
            out[..., j] = (A[..., self.i1[sl]] * B[..., self.i2[sl]] * self.val[sl]).sum(axis=-1)
        return out


def draw_sketch(rng, k1, k2, rank_bound, sparsity):
    D = k1 * k2
    r = int(rank_bound)
    if sparsity == 'lin':                                    # low_rank_calculations.py:104-127
        if r > D:
            raise ValueError("rank_bound exceeds the number of coordinate pairs")
        sel = rng.choice(D, size=r, replace=False)           # the first r of a shuffle of all pairs (:120-122), without shuffling all D
        i1, i2 = sel % k1, sel // k1                         # combinations[...] = (idx1 over k1 fastest, idx2)
        sign = np.where(rng.random(r) <= 0.5, 1.0, -1.0)
        return Sketch(k1, k2, r, np.arange(r + 1), i1, i2, sign)
    if sparsity == 'sqrt':                                   # :172-175
        s = np.sqrt(float(D))
    elif sparsity == 'log':
        s = float(D) / np.log(float(D))
    else:
        raise ValueError("Unknown sparsity argument %s. Possible values are 'sqrt', 'log', 'lin'" % sparsity)
    # R (D, r): every entry N(0, 1) with probability 1/s, else 0 (:144-149, :177), times sqrt(s / r) (:192).  Drawn as what it is --
    # a Binomial(D r, 1/s) number of entries at uniformly distributed distinct positions -- instead of D r uniforms and normals
    # (the dense draw was a third of the 12 ms one evaluation's random objects took at num_components = 50).
    total = D * r
    nnz = int(rng.binomial(total, min(1.0, 1.0 / s)))
    pos = rng.choice(total, size=nnz, replace=False, shuffle=False) if nnz else np.zeros(0, dtype=np.int64)
    key = np.sort((pos % r) * D + pos // r)                  # by output column, rows ascending within a column (one sort of a combined
    col, row = key // D, key % D                             # key: np.lexsort took more than half of a draw)
    val = rng.standard_normal(nnz) * np.sqrt(s / r)
    colptr = np.concatenate(([0], np.cumsum(np.bincount(col, minlength=r)))) if nnz else np.zeros(r + 1, dtype=np.int64)
    return Sketch(k1, k2, r, colptr, row % k1, row // k1, val)


def draw_level_sketches(rng, num_levels, num_components, rank_bound, sparsity):
    """One sketch per level 2..M: level 2 contracts (c, c), later levels (c, rank_bound)."""
    out, k2 = [], num_components
    for _ in range(2, num_levels + 1):
        out.append(draw_sketch(rng, num_components, k2, rank_bound, sparsity))
        k2 = rank_bound
    return out


def draw_landmarks(rng, points, num_components):
    """low_rank_calculations.py:12-20, :47-48: rows drawn without replacement."""
    n = points.shape[0]
    if num_components > n:
        raise ValueError("num_components exceeds the number of available points")
    return points[rng.choice(n, size=num_components, replace=False)]
