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

// Linear base kernel with increments (kernels.py:328-330): kappa(z1, x) - kappa(z0, x) = <z1 - z0, x>, so the pair of points
// collapses to its difference on the way in, and the gradient fans out as (-g, +g) on the way out.
__global__ void grad_pad_diff_rows_kernel(const double* __restrict__ Z, double* __restrict__ ZP, int64_t rows, int d, int DP) {
    const int64_t total = rows * DP;      // rows = lt * T (each holding two points in Z)
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int f = int(e % DP);
        const int64_t r = e / DP;
        ZP[e] = f < d ? Z[(2 * r + 1) * d + f] - Z[(2 * r) * d + f] : 0.0;
    }
}
__global__ void grad_unpad_pm_rows_kernel(const double* __restrict__ ZP, double* __restrict__ Z, int64_t rows, int d, int DP) {
    const int64_t total = rows * d;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int f = int(e % d);
        const int64_t r = e / d;
        const double g = ZP[r * DP + f];
        Z[(2 * r) * d + f] = -g;
        Z[(2 * r + 1) * d + f] = g;
    }
}

// Weighted level sums  S[t][n] = sum_m fac[n][m] level_m[t][n]  (gpsig_tens_vs_seq_weighted) through the level primitives, for the shapes
// the tile kernels are not built for: upstream gradient of the levels, and the gradient with respect to the factors.
__global__ void weighted_upstream_levels_kernel(const double* __restrict__ G, const double* __restrict__ fac, int M1, int64_t T, int64_t N,
                                                double* __restrict__ Glev) {
    const int64_t total = int64_t(M1) * T * N;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t n = e % N, t = (e / N) % T;
        const int m = int(e / (N * T));
        Glev[e] = G[t * N + n] * fac[n * M1 + m];
    }
}
__global__ void weighted_gfac_kernel(const double* __restrict__ G, const double* __restrict__ lev, int M1, int64_t T, int64_t N,
                                     double* __restrict__ gfac) {
    const int64_t total = N * M1;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t n = e % N;
        const int m = int(e / N);
        double s = 0.0;
        for (int64_t t = 0; t < T; ++t) s = fma(G[t * N + n], lev[(int64_t(m) * T + t) * N + n], s);
        gfac[n * M1 + m] = s;
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
    if (A.order > 1) {
        P.forward_ho();          // higher-order chains: every level's backward pass follows its forward pass
    } else {
        P.forward();
        P.backward();
    }
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


// Tensor-lane variant of the tensor-vs-sequence gradient: lane = inducing tensor (64 per wavefront), the sequences of a run
// are streamed one after the other through LDS (broadcast reads), the level's components of the 64 tensors sit in LDS
// as well, the per-lane accumulators of d/dz live in registers for the whole run, and the gradient of an observation is
// summed over the 64 tensors through an LDS transpose before ONE 8*DP-byte atomic per (level, sequence, time point).
struct TvsLaneTGradArgs {
    const double* z;        // scaled components padded to DP: (lt, T, E, DP)
    const double* X;        // scaled sequences, user layout (N, L, d)
    double* gz;             // (lt, T, E, DP), accumulated
    double* gX;             // (N, L, d), accumulated
    int T, N, L, d, M, kind, diff;
    double p0, p1;
    const double* G; int64_t gm, gt, gn;
    int nrun;               // sequences per workgroup
    double* gbase;
};

// ZREG: the level's components live in registers (affordable when E == 1) instead of LDS
template <int DP, int MMAX, int E, bool ZREG, bool PAIRED = false>
struct TvsLaneTIO {
    static constexpr int ZP = DP + 2;      // row stride: 16 consecutive lanes hit 16 different 16-byte bank groups
    const double* zs; const double* xs; const double* xsq; double* red;
    const TvsLaneTGradArgs& A;
    int lane, n;
    bool valid;                // this lane holds a tensor
    double zr[ZREG ? MMAX : 1][E][DP];
    double zn[ZREG ? MMAX : 1][E];   // squared norms of the level's components (kept only next to register-resident components)
    int flip;                  // which of the two reduction buffers the next emit uses
    // PAIRED: lanes 2t and 2t+1 hold the two points of incremental tensor t (kernels.py:328-330: kappa(z1, x) - kappa(z0, x))
    __device__ __forceinline__ double sign() const { return (PAIRED && (lane & 1) == 0) ? -1.0 : 1.0; }
    __device__ __forceinline__ double combine(double k) const {
        if constexpr (!PAIRED) return k;
        else {
            int lo = __double2loint(k), hi = __double2hiint(k);
            lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true);      // quad_perm:[1,0,3,2]
            hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true);
            return k + __hiloint2double(hi, lo);
        }
    }
    __device__ __forceinline__ double z(int k, int e, int f) const {
        if constexpr (ZREG) return zr[k][e][f];
        else return zs[((k * E + e) * 64 + lane) * ZP + f];
    }
    __device__ __forceinline__ double zsq(int k, int e) const {
        if constexpr (ZREG) return zn[k][e];
        else {
            double s = 0.0;
#pragma unroll
            for (int f = 0; f < DP; ++f) s = fma(z(k, e, f), z(k, e, f), s);
            return s;
        }
    }
    __device__ __forceinline__ double load_x(int tt, double (&v)[DP]) const {
#pragma unroll
        for (int f = 0; f < DP; ++f) v[f] = xs[tt * DP + f];
        return xsq[tt];
    }
    // sum of gx over the 64 tensors of the wavefront -> one atomic of DP doubles.  Two buffers: one barrier per call.
    __device__ __forceinline__ void emit_gx(int tt, const double (&gx)[DP]) {
        double* rb = red + flip * 64 * ZP;
        flip ^= 1;
#pragma unroll
        for (int f = 0; f < DP; ++f) rb[lane * ZP + f] = valid ? gx[f] : 0.0;
        __syncthreads();
        const int f = lane % DP, part = lane / DP;
        double sacc = 0.0;
#pragma unroll
        for (int r = 0; r < DP; ++r) sacc += rb[(part * DP + r) * ZP + f];
#pragma unroll
        for (int o = DP; o < 64; o <<= 1) sacc += __shfl_xor(sacc, o, 64);
        if (lane < DP && lane < A.d) atomicAdd(&A.gX[(int64_t(n) * A.L + tt) * A.d + lane], sacc);
    }
};

// grid (ceil(T / 64), runs[, levels]); block 64; dynamic LDS: ((ZREG ? 0 : MMAX * E * 64 * (DP + 2)) + L * DP + L + 2 * 64 * (DP + 2)) doubles
// PAIRED (E == 1 in the template, two points per tensor in the data): 32 tensors per wavefront, lane 2t + e holds point e
template <int DP, int MMAX, int E, int KIND, bool ZREG, bool PAIRED = false>
__global__ void __launch_bounds__(64) tvs_grad_lanet_kernel(const TvsLaneTGradArgs A) {
    extern __shared__ double tvs_sm[];
    constexpr int ZP = DP + 2;
    double* zs = tvs_sm;
    double* xs = zs + (ZREG ? 0 : MMAX * E * 64 * ZP);
    double* xsq = xs + A.L * DP;
    double* red = xsq + A.L;
    const int lane = threadIdx.x;
    const int t = PAIRED ? blockIdx.x * 32 + lane / 2 : blockIdx.x * 64 + lane;
    constexpr int EZ = PAIRED ? 2 : E;               // points per tensor in A.z / A.gz
    const int pe = PAIRED ? (lane & 1) : 0;          // which of them this lane holds
    const bool valid = t < A.T;
    const int n0 = blockIdx.y * A.nrun, n1 = (n0 + A.nrun < A.N) ? n0 + A.nrun : A.N;
    const int R = A.diff ? A.L - 1 : A.L;
    TvsLaneTIO<DP, MMAX, E, ZREG, PAIRED> io{zs, xs, xsq, red, A, lane, 0, valid, {}, {}, 0};
    int k0 = 0;
    double gp0 = 0.0;
    for (int i = 1; i <= A.M; ++i) {
        if (gridDim.z > 1 && i != int(blockIdx.z) + 1) { k0 += i; continue; }      // small problems: one level per workgroup
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MMAX; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                double nrm = 0.0;
#pragma unroll
                for (int f = 0; f < DP; ++f) {
                    // lanes without a tensor get a harmless finite point
                    const double v = (j < i && valid) ? A.z[((int64_t(k0 + j) * A.T + t) * EZ + (PAIRED ? pe : e)) * DP + f] : 1.0;
                    if constexpr (ZREG) io.zr[j][e][f] = v;
                    else if (j < i) zs[((j * E + e) * 64 + lane) * ZP + f] = v;
                    nrm = fma(v, v, nrm);
                }
                if constexpr (ZREG) io.zn[j][e] = nrm;
            }
        double gzacc[MMAX][E][DP];
#pragma unroll
        for (int j = 0; j < MMAX; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e)
#pragma unroll
                for (int f = 0; f < DP; ++f) gzacc[j][e][f] = 0.0;
        for (int n = n0; n < n1; ++n) {
            __syncthreads();
            for (int e = lane; e < A.L * DP; e += 64) {
                const int q = e / DP, f = e % DP;
                xs[e] = f < A.d ? A.X[(int64_t(n) * A.L + q) * A.d + f] : 0.0;
            }
            __syncthreads();
            for (int q = lane; q < A.L; q += 64) {
                double sq = 0.0;
#pragma unroll
                for (int f = 0; f < DP; ++f) sq = fma(xs[q * DP + f], xs[q * DP + f], sq);
                xsq[q] = sq;
            }
            __syncthreads();
            io.n = n;
            const double c = valid ? A.G[i * A.gm + t * A.gt + n * A.gn] : 0.0;
            switch (i) {
                case 1: tvs_level_grad<DP, MMAX, E, KIND, 1>(io, i, 0, R, A.diff != 0, A.kind, A.p0, A.p1, c, gzacc, gp0); break;
                case 2: tvs_level_grad<DP, MMAX, E, KIND, 2>(io, i, 0, R, A.diff != 0, A.kind, A.p0, A.p1, c, gzacc, gp0); break;
                case 3: tvs_level_grad<DP, MMAX, E, KIND, 3>(io, i, 0, R, A.diff != 0, A.kind, A.p0, A.p1, c, gzacc, gp0); break;
                default: tvs_level_grad<DP, MMAX, E, KIND, MMAX>(io, i, 0, R, A.diff != 0, A.kind, A.p0, A.p1, c, gzacc, gp0); break;
            }
        }
        if (valid) {
#pragma unroll
            for (int j = 0; j < MMAX; ++j)
                if (j < i) {
#pragma unroll
                    for (int e = 0; e < E; ++e)
#pragma unroll
                        for (int f = 0; f < DP; ++f) atomicAdd(&A.gz[((int64_t(k0 + j) * A.T + t) * EZ + (PAIRED ? pe : e)) * DP + f], gzacc[j][e][f]);
                }
        }
        k0 += i;
    }
    if (A.gbase) grad_add(&A.gbase[0], gp0, true, valid);
}

// row-owned tensor-vs-tensor gradient: grid (ceil(T / 64), slices, components); block 64: lanes = t
template <int DP, int E>
__global__ void __launch_bounds__(64) tens_row_grad_kernel(const TensGradArgs A) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    const bool valid = t < A.T;
    TensRowGrad<DP, E>(A, valid ? t : 0, valid).run(blockIdx.y, gridDim.y, gridDim.z > 1 ? int(blockIdx.z) : -1);
}

// gz[i] = sum over slices of part[s * n + i]
static __global__ void tens_row_reduce_kernel(const double* __restrict__ part, double* __restrict__ gz, int nslices, int64_t n) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // eight loads in flight per thread
    int k = 0;
    for (; k + 8 <= nslices; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += part[int64_t(k + u) * n + i];
    }
    for (; k < nslices; ++k) s[0] += part[int64_t(k) * n + i];
    gz[i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
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
