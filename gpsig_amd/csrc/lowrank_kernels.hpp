// lowrank_kernels.hpp -- kernels of the low-rank signature-kernel algorithms
// (gpsig/low_rank_calculations.py:26-193, gpsig/signature_algs.py:162-222, gpsig/kernels.py:239-311).
//
// Everything random (landmarks, sparse projections) is drawn on the host and arrives as plain arrays
// (gpsig_amd/low_rank.py); these kernels are deterministic.  The one dense contraction of the path -- the
// Gram of the low-rank factors, (N1, F) x (N2, F)^T -- runs on the fp64 matrix cores (v_mfma_f64_16x16x4).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aux_kernels.hpp"
#include "seq_core.hpp"

namespace gpsig {

// Scaled points by row index: out[r][fe] = x~ of flat point idx[r] of X (N, L, d)   (landmark gather).
template <typename T>
__global__ void lr_gather_points_kernel(const T* __restrict__ X, int L, ScaleParams P, const int64_t* __restrict__ idx, int64_t R,
                                        T* __restrict__ out) {
    const int d_eff = P.d_eff();
    const int64_t total = R * d_eff;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int fe = int(i % d_eff);
        const int64_t r = i / d_eff;
        const int64_t n = idx[r] / L;
        const int t = int(idx[r] % L);
        out[i] = scaled_point<T>(X + n * int64_t(L) * P.d_in, L, t, fe, P);
    }
}

// Base-kernel matrix of already scaled points: out[a][b] = kappa(A[a], B[b]); A (na, d), B (nb, d).
template <typename T>
__global__ void base_kernel_matrix_kernel(const T* __restrict__ A, const T* __restrict__ B, int64_t na, int64_t nb, int d, int kind,
                                          T p0, T p1, const double* __restrict__ spec, T* __restrict__ out) {
    const int64_t total = na * nb;
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
        const int64_t a = i / nb, b = i % nb;
        if (kind == BASE_SPECTRAL) {
            out[i] = spectral_eval<T>(spec, int(p0), int(p1), d, [&](int f) { return A[a * d + f]; }, [&](int f) { return B[b * d + f]; });
            continue;
        }
        T ip = T(0), as = T(0), bs = T(0);
        for (int f = 0; f < d; ++f) {
            const T x = A[a * d + f], y = B[b * d + f];
            ip = fma(x, y, ip); as = fma(x, x, as); bs = fma(y, y, bs);
        }
        out[i] = base_eval<T>(kind, ip, as, bs, p0, p1);
    }
}

// Nystrom cross matrix of sequences (low_rank_calculations.py:59): out[(n*L + t)][i] = kappa(x~[n][t], S[i]).
template <typename T>
__global__ void lr_seq_cross_kernel(const T* __restrict__ X, int64_t N, int L, ScaleParams P, const T* __restrict__ S, int c, int kind,
                                    T p0, T p1, T* __restrict__ out) {
    const int d_eff = P.d_eff();
    const int64_t total = N * L * c;
    for (int64_t q = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; q < total; q += int64_t(gridDim.x) * blockDim.x) {
        const int i = int(q % c);
        const int64_t pt = q / c;
        const int64_t n = pt / L;
        const int t = int(pt % L);
        const T* Xn = X + n * int64_t(L) * P.d_in;
        T ip = T(0), xs = T(0), ss = T(0);
        for (int fe = 0; fe < d_eff; ++fe) {
            const T x = scaled_point<T>(Xn, L, t, fe, P), y = S[i * d_eff + fe];
            ip = fma(x, y, ip); xs = fma(x, x, xs); ss = fma(y, y, ss);
        }
        out[q] = base_eval<T>(kind, ip, xs, ss, p0, p1);
    }
}

// The same for scaled tensor components: Z (lt, T, E, d') as the caller gives it -> out[((k*T + t)*E + e)][i].
template <typename T>
__global__ void lr_tens_cross_kernel(const T* __restrict__ Z, int64_t rows, ScaleParams P, const T* __restrict__ S, int c, int kind,
                                     T p0, T p1, T* __restrict__ out) {
    const int d_eff = P.d_eff();
    const int64_t total = rows * c;
    for (int64_t q = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; q < total; q += int64_t(gridDim.x) * blockDim.x) {
        const int i = int(q % c);
        const int64_t r = q / c;
        T ip = T(0), xs = T(0), ss = T(0);
        for (int fe = 0; fe < d_eff; ++fe) {
            const int lag = fe / P.d_in, f = fe - lag * P.d_in;
            T x = Z[r * d_eff + fe];
            if (P.has_ls) {                                  // kernels.py:374-379 / :391-395
                x = x / T(P.lsv(f));
                if (P.num_lags > 0) x = x * T(P.gamma[lag]);
            }
            const T y = S[i * d_eff + fe];
            ip = fma(x, y, ip); xs = fma(x, x, xs); ss = fma(y, y, ss);
        }
        out[q] = base_eval<T>(kind, ip, xs, ss, p0, p1);
    }
}

// U[n][t][j] = F[n][t+1][j] - F[n][t][j]   (signature_algs.py:180), rows_out = L-1;  or a plain copy (difference=False)
template <typename T>
__global__ void lr_time_diff_kernel(const T* __restrict__ F, int64_t N, int L, int c, int difference, T* __restrict__ U) {
    const int lo = difference ? L - 1 : L;
    const int64_t total = N * lo * c;
    for (int64_t q = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; q < total; q += int64_t(gridDim.x) * blockDim.x) {
        const int j = int(q % c);
        const int t = int((q / c) % lo);
        const int64_t n = q / (int64_t(c) * lo);
        const T* f = F + (n * L + t) * int64_t(c) + j;
        U[q] = difference ? f[c] - f[0] : f[0];
    }
}

// In-place exclusive cumulative sum over time (signature_algs.py:186) of P (N, l, k); also Phi[n][off + j] = total.
template <typename T>
__global__ void lr_excumsum_kernel(T* __restrict__ P, int64_t N, int l, int k, T* __restrict__ Phi, int64_t phi_stride, int phi_off,
                                   int write_phi) {
    const int64_t total = N * k;
    for (int64_t q = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; q < total; q += int64_t(gridDim.x) * blockDim.x) {
        const int j = int(q % k);
        const int64_t n = q / k;
        T run = T(0);
        for (int t = 0; t < l; ++t) {
            T* p = P + (n * l + t) * int64_t(k) + j;
            const T v = *p;
            *p = run;
            run += v;
        }
        if (write_phi) Phi[n * phi_stride + phi_off + j] = run;      // sum over time (signature_algs.py:182, :191)
    }
}

// Phi[n][off + j] = sum_t P[n][t][j] without touching P.
template <typename T>
__global__ void lr_timesum_kernel(const T* __restrict__ P, int64_t N, int l, int k, T* __restrict__ Phi, int64_t phi_stride, int phi_off) {
    const int64_t total = N * k;
    for (int64_t q = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; q < total; q += int64_t(gridDim.x) * blockDim.x) {
        const int j = int(q % k);
        const int64_t n = q / k;
        T run = T(0);
        for (int t = 0; t < l; ++t) run += P[(n * l + t) * int64_t(k) + j];
        Phi[n * phi_stride + phi_off + j] = run;
    }
}

// Randomised low-rank Hadamard product (low_rank_calculations.py:64-193) with a host-drawn sparse projection stored by
// output column: out[row][j] = sum_{e in col j} val[e] * A[row][i1[e]] * B[row][i2[e]].   A (rows, k1), B (rows, k2).
template <typename T>
__global__ void lr_sketch_kernel(const T* __restrict__ A, int64_t a_stride, const T* __restrict__ B, int64_t b_stride, int64_t rows,
                                 int r, const int32_t* __restrict__ colptr, const int32_t* __restrict__ i1,
                                 const int32_t* __restrict__ i2, const double* __restrict__ val, T* __restrict__ out, int64_t out_stride) {
    const int64_t total = rows * r;
    for (int64_t q = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; q < total; q += int64_t(gridDim.x) * blockDim.x) {
        const int j = int(q % r);
        const int64_t row = q / r;
        const T* a = A + row * a_stride;
        const T* b = B + row * b_stride;
        T acc = T(0);
        for (int e = colptr[j]; e < colptr[j + 1]; ++e) acc = fma(T(val[e]) * a[i1[e]], b[i2[e]], acc);
        out[row * out_stride + j] = acc;
    }
}

// Per-row, per-level block norms of a factor matrix Phi (N, F) whose level m occupies columns [off[m], off[m+1]):
// fac[n][m] = num[m] / sqrt(|Phi_m[n]|^2 + jitter)   (normalise != 0)   or   num[m].
template <typename T>
__global__ void lr_level_factors_kernel(const T* __restrict__ Phi, int64_t N, int64_t F, int M1, const int32_t* __restrict__ off,
                                        const double* __restrict__ num, double jitter, int normalise, T* __restrict__ fac) {
    const int64_t total = N * M1;
    for (int64_t q = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; q < total; q += int64_t(gridDim.x) * blockDim.x) {
        const int m = int(q % M1);
        const int64_t n = q / M1;
        T ss = T(0);
        for (int j = off[m]; j < off[m + 1]; ++j) { const T v = Phi[n * F + j]; ss = fma(v, v, ss); }
        const T nm = num ? T(num[m]) : T(1);
        fac[q] = normalise ? nm / sqrt(ss + T(jitter)) : nm;
    }
}

// out[n][j] = Phi[n][j] * fac[n][level(j)]; optionally restricted to one level's block (level >= 0) written densely.
template <typename T>
__global__ void lr_scale_factors_kernel(const T* __restrict__ Phi, int64_t N, int64_t F, int M1, const int32_t* __restrict__ off,
                                        const T* __restrict__ fac, int level, T* __restrict__ out, int64_t out_stride) {
    const int jlo = level >= 0 ? off[level] : 0, jhi = level >= 0 ? off[level + 1] : int(F);
    const int w = jhi - jlo;
    const int64_t total = N * w;
    for (int64_t q = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; q < total; q += int64_t(gridDim.x) * blockDim.x) {
        const int j = jlo + int(q % w);
        const int64_t n = q / w;
        int m = 0;
        while (m + 1 < M1 && j >= off[m + 1]) ++m;
        out[n * out_stride + (j - jlo)] = Phi[n * F + j] * fac[n * M1 + m];
    }
}

// diag term of the symmetric normalised Gram: out[i][i] += sum over the selected levels of jitter * fa[i][m] * fb[i][m]
template <typename T>
__global__ void lr_add_jitter_diag_kernel(T* __restrict__ out, int64_t N, int M1, const T* __restrict__ fa, const T* __restrict__ fb,
                                          double jitter, int level) {
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < N; i += int64_t(gridDim.x) * blockDim.x) {
        T acc = T(0);
        for (int m = 0; m < M1; ++m)
            if (level < 0 || m == level) acc += T(jitter) * fa[i * M1 + m] * fb[i * M1 + m];
        out[i * N + i] += acc;
    }
}

// p[n * stride] = v
template <typename T>
__global__ void fill_strided_kernel(T* __restrict__ p, int64_t N, int64_t stride, T v) {
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < N; i += int64_t(gridDim.x) * blockDim.x) p[i * stride] = v;
}

// dst[n][off + j] = src[n][j]
template <typename T>
__global__ void copy_block_kernel(const T* __restrict__ src, int64_t N, int w, int64_t src_stride, T* __restrict__ dst, int64_t dst_stride, int off) {
    const int64_t total = N * w;
    for (int64_t q = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; q < total; q += int64_t(gridDim.x) * blockDim.x) {
        const int j = int(q % w);
        const int64_t n = q / w;
        dst[n * dst_stride + off + j] = src[n * src_stride + j];
    }
}

// out[m][n] (levels) or out[n] (sum) = w[m] * |Phi_m[n]|^2        (kernels.py:499-510)
template <typename T>
__global__ void lr_level_diag_kernel(const T* __restrict__ Phi, int64_t N, int64_t F, int M1, const int32_t* __restrict__ off,
                                     const double* __restrict__ w, int levels, T* __restrict__ out) {
    if (levels) {
        const int64_t total = N * M1;
        for (int64_t q = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; q < total; q += int64_t(gridDim.x) * blockDim.x) {
            const int64_t n = q % N;
            const int m = int(q / N);
            T ss = T(0);
            for (int j = off[m]; j < off[m + 1]; ++j) { const T v = Phi[n * F + j]; ss = fma(v, v, ss); }
            out[q] = ss * T(w[m]);
        }
    } else {
        for (int64_t n = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; n < N; n += int64_t(gridDim.x) * blockDim.x) {
            T acc = T(0);
            for (int m = 0; m < M1; ++m) {
                T ss = T(0);
                for (int j = off[m]; j < off[m + 1]; ++j) { const T v = Phi[n * F + j]; ss = fma(v, v, ss); }
                acc += ss * T(w[m]);
            }
            out[n] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// C (N1, N2) = A (N1, K) * B (N2, K)^T on the fp64 matrix cores.  One wavefront per 32 x 32 output tile (2 x 2
// MFMA tiles of 16 x 16, K advanced 4 at a time).  v_mfma_f64_16x16x4_f64 operand layout (cdna_hip_programming.md):
// A operand: lane l holds A[i = l & 15][k = l >> 4]; B operand: lane l holds B[k = l >> 4][j = l & 15]; the four
// results of lane l are D[(l >> 4) + 4 r][l & 15], r = 0..3.  K is small here (a few hundred): the kernel is bound by
// writing C, operands come straight from L2.
typedef double mfma_f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gemm_abt_f64_mfma_kernel(const double* __restrict__ A, const double* __restrict__ B, int64_t N1,
                                                               int64_t N2, int K, int64_t lda, int64_t ldb, double* __restrict__ C,
                                                               int64_t ldc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile_i = (int64_t(blockIdx.y) * 2 + (wave >> 1)) * 32;     // 64 x 64 per block, 32 x 32 per wave
    const int64_t tile_j = (int64_t(blockIdx.x) * 2 + (wave & 1)) * 32;
    if (tile_i >= N1 || tile_j >= N2) return;
    const int li = lane & 15, lk = lane >> 4;
    mfma_f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma_f64x4{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + lk;
        double av[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int64_t i = tile_i + 16 * a + li;
            av[a] = (i < N1 && k < K) ? A[i * lda + k] : 0.0;
            const int64_t j = tile_j + 16 * a + li;
            bv[a] = (j < N2 && k < K) ? B[j * ldb + k] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = tile_i + 16 * a + lk + 4 * r, j = tile_j + 16 * b + li;
                if (i < N1 && j < N2) C[i * ldc + j] = acc[a][b][r];
            }
}

// The same product in 128 x 128 tiles per workgroup of four wavefronts (64 x 64 = 4 x 4 MFMA tiles each), operands staged
// through LDS: a k-slab of 16 columns of both operands is fetched coalesced (every thread 8 consecutive doubles of one row)
// into registers while the previous slab is multiplied, then written to LDS rows padded to 17 doubles (the 16 rows x 4 k a
// fragment read touches fall on different banks).  gemm_abt_f64_mfma_kernel above reads every fragment straight from L2 at a
// stride of one row per lane: 23-26 TFLOP/s at the low-rank Grams' shapes (K = 201 .. 251); this one is bound by the matrix
// pipe once K is a few slabs deep.
constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 16, GEMM_LDK = GEMM_BK + 1;
__global__ __launch_bounds__(256, 2) void gemm_abt_f64_tiled_kernel(const double* __restrict__ A, const double* __restrict__ B, int64_t N1,
                                                                   int64_t N2, int K, int64_t lda, int64_t ldb, double* __restrict__ C,
                                                                   int64_t ldc) {
    __shared__ double As[GEMM_BM * GEMM_LDK];
    __shared__ double Bs[GEMM_BN * GEMM_LDK];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const int64_t tile_i = int64_t(blockIdx.y) * GEMM_BM, tile_j = int64_t(blockIdx.x) * GEMM_BN;
    // staging: thread t fetches columns [8 h, 8 h + 8) of row t >> 1 of the slab, h = t & 1
    const int srow = tid >> 1, scol = (tid & 1) * 8;
    const int64_t ai = tile_i + srow, bj = tile_j + srow;
    const double* arow = A + (ai < N1 ? ai : 0) * lda;
    const double* brow = B + (bj < N2 ? bj : 0) * ldb;
    const bool aok = ai < N1, bok = bj < N2;
    double pa[8], pb[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + scol + e;
            pa[e] = (aok && k < K) ? arow[k] : 0.0;
            pb[e] = (bok && k < K) ? brow[k] : 0.0;
        }
    };
    mfma_f64x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = mfma_f64x4{0.0, 0.0, 0.0, 0.0};
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += GEMM_BK) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            As[srow * GEMM_LDK + scol + e] = pa[e];
            Bs[srow * GEMM_LDK + scol + e] = pb[e];
        }
        __syncthreads();
        if (k0 + GEMM_BK < K) fetch(k0 + GEMM_BK);          // in flight while this slab is multiplied
#pragma unroll
        for (int kk = 0; kk < GEMM_BK; kk += 4) {
            double av[4], bv[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                av[m] = As[(wr * 64 + m * 16 + li) * GEMM_LDK + kk + lk];
                bv[m] = Bs[(wc * 64 + m * 16 + li) * GEMM_LDK + kk + lk];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = tile_i + wr * 64 + m * 16 + lk + 4 * r, j = tile_j + wc * 64 + n * 16 + li;
                if (i < N1 && j < N2) C[i * ldc + j] = acc[m][n][r];
            }
}

}  // namespace gpsig
