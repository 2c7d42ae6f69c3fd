// grad_kernels.hpp -- gfx950 kernels of the gradient path: one pair per thread (grad_core.hpp), 64 pairs per
// workgroup = one wavefront, lattice state in an HBM scratch array laid out so that a wavefront's accesses are
// 512 contiguous bytes.  This is the simple, storage-based formulation (what reverse-mode autodiff of the
// reference's graph does, minus the 4-D intermediates); its cost is HBM traffic on the scratch lattice.
#pragma once

#include <hip/hip_runtime.h>

#include "grad_core.hpp"

namespace gpsig {

// (N, L, d) row-major -> time-major [(t * DP + f) * stride + i], features zero-padded to DP, sequences i >= N zero
__global__ void grad_to_timemajor_kernel(const double* __restrict__ X, double* __restrict__ XT, int N, int L, int d, int DP, int64_t stride) {
    const int64_t total = int64_t(L) * DP * stride;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t i = e % stride;
        const int64_t tf = e / stride;
        const int f = int(tf % DP), t = int(tf / DP);
        XT[e] = (i < N && f < d) ? X[(i * L + t) * d + f] : 0.0;
    }
}

// gX[i][t][f] (+)= gXT[(t * DP + f) * stride + i]
__global__ void grad_from_timemajor_kernel(const double* __restrict__ gXT, double* __restrict__ gX, int N, int L, int d, int DP, int64_t stride,
                                           int accumulate) {
    const int64_t total = int64_t(N) * L * d;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int f = int(e % d);
        const int64_t it = e / d;
        const int t = int(it % L);
        const int64_t i = it / L;
        const double v = gXT[(int64_t(t) * DP + f) * stride + i];
        gX[e] = accumulate ? gX[e] + v : v;
    }
}

// rows of d features <-> rows of DP features
__global__ void grad_pad_rows_kernel(const double* __restrict__ Z, double* __restrict__ ZP, int64_t rows, int d, int DP) {
    const int64_t total = rows * DP;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int f = int(e % DP);
        ZP[e] = f < d ? Z[(e / DP) * d + f] : 0.0;
    }
}
__global__ void grad_unpad_rows_kernel(const double* __restrict__ ZP, double* __restrict__ Z, int64_t rows, int d, int DP, int accumulate) {
    const int64_t total = rows * d;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const double v = ZP[(e / d) * DP + e % d];
        Z[e] = accumulate ? Z[e] + v : v;
    }
}

// grid (ceil(N1 / 64), nj or 1); block 64
template <int DP>
__global__ void __launch_bounds__(64) seq_pair_grad_kernel(const SeqGradArgs A) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = A.diag ? i : A.j0 + int(blockIdx.y);
    const int64_t pidx = (int64_t(blockIdx.y) * gridDim.x + blockIdx.x) * 64 + threadIdx.x;
    const bool valid = i < A.N1;
    SeqPairGrad<DP> P(A, i, j, pidx, valid);
    P.forward();
    P.backward();
    P.contract();
}

// grid (ceil(N / 64), nt); block 64
template <int DP>
__global__ void __launch_bounds__(64) tvs_pair_grad_kernel(const TvsGradArgs A) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    const int t = A.t0 + int(blockIdx.y);
    const int64_t pidx = (int64_t(blockIdx.y) * gridDim.x + blockIdx.x) * 64 + threadIdx.x;
    TvsPairGrad<DP> P(A, t, n, pidx, n < A.N);
    P.forward();
    P.backward();
    P.contract();
}

// scratch-free variant: grid (ceil(N / 64), T); block 64
template <int DP, int MMAX, int E>
__global__ void __launch_bounds__(64) tvs_pair_grad_fused_kernel(const TvsGradArgs A) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    const int t = blockIdx.y;
    TvsPairGradFused<DP, MMAX, E> P(A, t, n < A.N ? n : 0, n < A.N);
    P.run();
}

// grid (ceil(T / 64), T); block 64: lanes = t2, blockIdx.y = t
template <int DP>
__global__ void __launch_bounds__(64) tens_pair_grad_kernel(const TensGradArgs A) {
    const int t2 = blockIdx.x * 64 + threadIdx.x;
    const int t = blockIdx.y;
    const bool valid = t2 < A.T;
    TensPairGrad<DP> P(A, t, valid ? t2 : 0, valid);
    P.run();
}

}  // namespace gpsig
