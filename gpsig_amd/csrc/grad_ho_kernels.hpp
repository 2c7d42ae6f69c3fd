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

}  // namespace gpsig
