// grad_ho_kernels.hpp -- reverse mode of the HIGHER-ORDER sequence-vs-sequence recursion
// (signature_kern_higher_order, gpsig/signature_algs.py:37-74), as lattice operations over a block of pairs.
//
// The reference differentiates the graph of :56-71 with TensorFlow's autodiff.  The same adjoint is taken here operation by
// operation on arrays shaped (pairs, R1, R2) -- one lattice per pair, per level m and per repeat-count pair (r, s):
//     R_1[0][0] = dM
//     R_m[0][0]     = dM * E_ab( sum_{r,s} R_{m-1}[r][s] )                      E_ab: exclusive cumsum along both axes   (:64)
//     R_m[0][j-1]   = dM * E_a ( sum_r R_{m-1}[r][j-2] ) / j                    E_a : along the x-time axis              (:66)
//     R_m[j-1][0]   = dM * E_b ( sum_s R_{m-1}[j-2][s] ) / j                    E_b : along the y-time axis              (:67)
//     R_m[j-1][k-1] = dM * R_{m-1}[j-2][k-2] / (j k)                                                                     (:69)
//     K_m = sum_ab sum_{r,s} R_m[r][s]                                                                                   (:71)
// With U_m[r][s] = dL/dR_m[r][s] (U_M = c_M everywhere) the backward pass walks the levels down: the multiplier of dM in
// each R_m[r][s] times U_m[r][s] accumulates Lam = dL/ddM, and U_{m-1} collects c_{m-1} plus the transposed operations
// (reverse exclusive cumsums of dM * U_m).  Lam then goes through lam_contract_kernel (grad_wave_kernel.hpp) like the
// first-order point kernels' -- the contraction with the base kernel's derivatives does not depend on the order.
// Three elementary kernels; the orchestration is seq_grad_ho in grad_api.hip.  This path is about coverage (training with
// order > 1, kernels.py:57), not speed: every operation is a separate pass over HBM-resident lattices.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "seq_core.hpp"

namespace gpsig {

constexpr int GRAD_HO_MAXLEV = 8;       // GRAD_MAX_LEVELS of grad_core.hpp

struct HoBlock {
    int64_t i0, ni, j0, nj;     // pairs (i, j), i in [i0, i0+ni), j in [j0, j0+nj); pair index p = (i - i0) * nj + (j - j0)
    int diag;                   // pairs (i, i): p = i - i0, nj == 1
};

// dM[p][a][b] (signature_algs.py:26 for the point modes; kappa itself without differences).  One thread per cell.
__global__ void ho_dm_kernel(const double* __restrict__ X, const double* __restrict__ Y, int L1, int L2, int d, int kind, int nodiff,
                             double p0, double p1, HoBlock B, int R1, int R2, double* __restrict__ dm) {
    const int64_t cells = int64_t(R1) * R2, total = B.ni * B.nj * cells;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t p = idx / cells;
        const int a = int((idx - p * cells) / R2), b = int(idx - p * cells - int64_t(a) * R2);
        const int64_t i = B.i0 + (B.diag ? p : p / B.nj), j = B.diag ? i : B.j0 + p % B.nj;
        auto kap = [&](int ta, int tb) {
            const double* x = X + (i * L1 + ta) * d;
            const double* y = Y + (j * L2 + tb) * d;
            double in = 0.0, xs = 0.0, ys = 0.0;
            for (int f = 0; f < d; ++f) { in = fma(x[f], y[f], in); xs = fma(x[f], x[f], xs); ys = fma(y[f], y[f], ys); }
            return base_eval<double>(kind, in, xs, ys, p0, p1);
        };
        dm[idx] = nodiff ? kap(a, b) : (kap(a + 1, b + 1) - kap(a, b + 1)) - (kap(a + 1, b) - kap(a, b));
    }
}

// The symmetric Gram's upstream gradient folded onto the pairs i <= j (the levels are symmetric functions of the two sequences):
// Gs[m][i][j] = G[m][i][j] + G[m][j][i] (j > i),  G[m][i][i] (j == i),  0 (j < i).
__global__ void ho_sym_upstream_kernel(const double* __restrict__ G, int64_t N, int M1, double* __restrict__ Gs) {
    const int64_t total = int64_t(M1) * N * N;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t j = idx % N, i = (idx / N) % N, m = idx / (N * N);
        Gs[idx] = j > i ? G[idx] + G[(m * N + j) * N + i] : (j == i ? G[idx] : 0.0);
    }
}

// dst = (acc ? dst : 0) + scale * E(src), E the EXCLUSIVE cumulative sum along axis 0 (a) or 1 (b), from the front or, with
// reverse, from the back (the transpose of the forward one).  One thread per lattice line; dst may alias src.
__global__ void ho_cumsum_kernel(const double* src, double* dst, int64_t npairs, int R1, int R2, int axis, int reverse, double scale, int acc) {
    const int lines = axis == 0 ? R2 : R1, len = axis == 0 ? R1 : R2;
    const int64_t stride = axis == 0 ? R2 : 1, total = npairs * lines;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t p = idx / lines;
        const int ln = int(idx - p * lines);
        const int64_t base = p * int64_t(R1) * R2 + (axis == 0 ? ln : int64_t(ln) * R2);
        double run = 0.0;
        for (int k = 0; k < len; ++k) {
            const int64_t o = base + int64_t(reverse ? len - 1 - k : k) * stride;
            const double v = src[o];
            dst[o] = (acc ? dst[o] : 0.0) + scale * run;
            run += v;
        }
    }
}

// dst = (acc ? dst : 0) + scale * A * B   (B == nullptr: scale * A)
__global__ void ho_mul_kernel(const double* A, const double* B, double* dst, int64_t n, double scale, int acc) {
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < n; idx += int64_t(gridDim.x) * blockDim.x) {
        const double v = scale * A[idx] * (B ? B[idx] : 1.0);
        dst[idx] = acc ? dst[idx] + v : v;
    }
}

// dst[p][:] = (acc ? dst : 0) + G[m * gm + i * gi + j * gj]: the upstream gradient of level m, one value per pair
__global__ void ho_bcast_kernel(const double* __restrict__ G, int64_t goff, int64_t gi, int64_t gj, HoBlock B, int64_t cells, double* dst, int acc) {
    const int64_t total = B.ni * B.nj * cells;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t p = idx / cells;
        const int64_t i = B.i0 + (B.diag ? p : p / B.nj), j = B.diag ? i : B.j0 + p % B.nj;
        const double c = G[goff + i * gi + j * gj];
        dst[idx] = acc ? dst[idx] + c : c;
    }
}

__global__ void fill_pairs_kernel(double* __restrict__ dst, int64_t n, double v) {
    for (int64_t e = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; e < n; e += int64_t(gridDim.x) * blockDim.x) dst[e] = v;
}

// out[pair] = (acc ? out[pair] : 0) + sum over the lattice cells of src[pair][:]  -- one wavefront per pair
__global__ void __launch_bounds__(64) ho_pairsum_kernel(const double* __restrict__ src, int64_t npairs, int64_t cells, double* __restrict__ out, int acc) {
    for (int64_t p = blockIdx.x; p < npairs; p += gridDim.x) {
        double s = 0.0;
        for (int64_t e = threadIdx.x; e < cells; e += 64) s += src[p * cells + e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (threadIdx.x == 0) out[p] = acc ? out[p] + s : s;
    }
}

// ---- tensor-vs-sequence chains from given component increments (signature_algs.py:101-127 after :114), order 1 -------------------
// m: (lt, R, P) -- component k = i(i-1)/2 + j of level i, time step tau, pair p (pairs fastest: a wavefront reads 512 contiguous bytes).
// One thread per pair.  out: (M+1, P).
__global__ void chain_levels_kernel(const double* __restrict__ m, int M, int64_t R, int64_t P, double* __restrict__ out) {
    for (int64_t p = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; p < P; p += int64_t(gridDim.x) * blockDim.x) {
        out[p] = 1.0;                                                              // :116
        int k0 = 0;
        for (int i = 1; i <= M; ++i) {
            double u[GRAD_HO_MAXLEV];
            for (int j = 0; j < i; ++j) u[j] = 0.0;
            for (int64_t t = 0; t < R; ++t) {
                for (int j = i - 1; j >= 1; --j) u[j] = fma(m[(int64_t(k0 + j) * R + t) * P + p], u[j - 1], u[j]);   // :120-124 (old values below)
                u[0] += m[(int64_t(k0) * R + t) * P + p];
            }
            out[int64_t(i) * P + p] = u[i - 1];                                    // :125
            k0 += i;
        }
    }
}
// gm[k][tau][p] = dL/dm[k][tau][p] for L = sum_i G[i][p] level_i[p]: forward totals, then the chains undone from the last step back
// (u_{j+1}[tau-1] = u_{j+1}[tau] - m_j[tau] u_j[tau-1]) with W_j = dL/du_j alongside, as tvs_grad_tile_kernel.hpp does on kappa's it evaluates.
__global__ void chain_levels_grad_kernel(const double* __restrict__ m, const double* __restrict__ G, int M, int64_t R, int64_t P,
                                         double* __restrict__ gm) {
    for (int64_t p = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; p < P; p += int64_t(gridDim.x) * blockDim.x) {
        int k0 = 0;
        for (int i = 1; i <= M; ++i) {
            double u[GRAD_HO_MAXLEV], w[GRAD_HO_MAXLEV];
            for (int j = 0; j < i; ++j) { u[j] = 0.0; w[j] = 0.0; }
            for (int64_t t = 0; t < R; ++t) {
                for (int j = i - 1; j >= 1; --j) u[j] = fma(m[(int64_t(k0 + j) * R + t) * P + p], u[j - 1], u[j]);
                u[0] += m[(int64_t(k0) * R + t) * P + p];
            }
            const double c = G[int64_t(i) * P + p];
            for (int64_t t = R - 1; t >= 0; --t) {
                double below = 1.0;
                for (int j = 0; j < i; ++j) {
                    const double mj = m[(int64_t(k0 + j) * R + t) * P + p];
                    const double wnext = (j == i - 1) ? c : w[j + 1];
                    gm[(int64_t(k0 + j) * R + t) * P + p] = below * wnext;
                    u[j] = fma(-mj, below, u[j]);
                    below = u[j];
                    if (j >= 1) w[j] = fma(mj, wnext, w[j]);
                }
            }
            k0 += i;
        }
    }
}

}  // namespace gpsig
