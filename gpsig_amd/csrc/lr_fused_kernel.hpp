// lr_fused_kernel.hpp -- the low-rank feature map of a batch of sequences in ONE kernel
// (gpsig/low_rank_calculations.py:59-60 Nystrom features, gpsig/signature_algs.py:166-193 signature_kern_first_order_lr_feature).
//
// The multi-pass form (lowrank_kernels.hpp, one elementwise kernel per reference op) moves every (N, L, c) intermediate through
// HBM about two dozen times; here a workgroup owns one sequence and keeps its three (width, L) arrays in LDS, so that HBM sees
// the sequence once (L * d values in) and its feature row once (F values out).
//
//   phase 0   scaled observations x~[t]                                                   (kernels.py:343-364, lags.py)
//   phase 1   kxs[t][i] = kappa(x~[t], S_i) against the c landmarks                        (low_rank_calculations.py:59)
//   phase 2   feat[t][j] = sum_i kxs[t][i] Wh[i][j];  U[t] = feat[t+1] - feat[t]           (:60, signature_algs.py:180)
//   level 1   Phi_1 = sum_t U[t]                                                           (:182)
//   level i   E = excumsum_t(P_{i-1});  P_i[t] = sketch_i(U[t], E[t]);  Phi_i = sum_t P_i  (:186-191)
//
// Layout: every array is stored [column][time] with an odd row stride, so that both access patterns are conflict-free:
// lane = time (phases 0-2 and the sketches: 64 consecutive doubles of one column) and lane = column (running sums over
// time: 64 addresses at an odd stride).  The sketch of a level -- the dominant cost, nnz multiply-adds per time step -- runs
// with lane = time: its entries (value, i1, i2) are the same for every lane, so they arrive through the scalar unit
// (one 16-byte s_load per entry) and the two operands are full-width LDS reads; the four wavefronts of the workgroup split
// the output columns.  Bound by LDS bandwidth: two 512-byte reads per entry and 64 time steps.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aux_kernels.hpp"
#include "lr_fused_args.hpp"
#include "seq_core.hpp"

namespace gpsig {

// THREADS: workgroup size (the LDS footprint of a sequence does not depend on it, so more wavefronts per workgroup are more
// wavefronts per CU to hide the scalar-load latency of the sketch entries behind); UNROLL: sketch entries per scalar-load batch.
// (These kernels switch over the base kernel at run time and sit on occupancy steps -- lr_seq_features_fused2_kernel at four wavefronts per SIMD =
// two workgroups per CU.  The repeated-squaring helper round 4 added to base_eval cost it ONE register and with it a workgroup per CU: BASELINE
// configs[2] in low-rank mode 2.8 -> 4.6 ms, unnoticed until the round's last bench.  The library pow is out of line everywhere since
// (seq_core.hpp: poly_pow_general); tests/test_abi.py reads the compiler's report for this kernel.)
template <int THREADS, int UNROLL>
__global__ __launch_bounds__(THREADS) void lr_seq_features_fused_kernel(LrFusedArgs A) {
    extern __shared__ double lr_lds[];
    const int lp = A.lp, c = A.c, r = A.r, L = A.L;
    double* const U = lr_lds;                               // [c][lp]
    double* bufA = U + size_t(c) * lp;                      // [rows_b][lp]
    double* bufB = bufA + size_t(A.rows_b) * lp;            // [rows_b][lp]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NW = THREADS / 64;
    const int d_eff = A.P.d_eff();
    const int l = A.difference ? L - 1 : L;                 // time steps of U
    const int nchunk = (L + 63) / 64;

    for (int64_t n = blockIdx.x; n < A.N; n += gridDim.x) {
        const double* Xn = A.X + n * int64_t(L) * A.P.d_in;
        double* phi = A.Phi + n * int64_t(A.F);
        // ---- phase 0: scaled observations, bufB[fe][t]
        for (int q = threadIdx.x; q < L * d_eff; q += THREADS) {
            const int t = q / d_eff, fe = q - t * d_eff;
            bufB[fe * lp + t] = scaled_point<double>(Xn, L, t, fe, A.P);
        }
        __syncthreads();
        // ---- phase 1: kxs, bufA[i][t]
        for (int ch = 0; ch < nchunk; ++ch) {
            const int t = ch * 64 + lane;
            if (t < L) {
                double xs = 0.0;
                for (int fe = 0; fe < d_eff; ++fe) { const double x = bufB[fe * lp + t]; xs = fma(x, x, xs); }
                for (int i = wave; i < c; i += NW) {
                    const lr_const_ptr<double> Si = lr_as_const(A.S) + size_t(i) * d_eff;
                    double ip = 0.0, ss = 0.0;
                    for (int fe = 0; fe < d_eff; ++fe) {
                        const double y = Si[fe];
                        ip = fma(bufB[fe * lp + t], y, ip);
                        ss = fma(y, y, ss);
                    }
                    bufA[i * lp + t] = base_eval<double>(A.kind, ip, xs, ss, A.p0, A.p1);
                }
            }
        }
        __syncthreads();
        // ---- phase 2: whitening, bufB[j][t] = sum_i bufA[i][t] * Wh[i][j]
        for (int ch = 0; ch < nchunk; ++ch) {
            const int t = ch * 64 + lane;
            if (t < L) {
                const lr_const_ptr<double> Wh = lr_as_const(A.Wh);
                for (int j = wave; j < c; j += NW) {
                    double acc = 0.0;
#pragma unroll 4
                    for (int i = 0; i < c; ++i) acc = fma(bufA[i * lp + t], Wh[size_t(i) * c + j], acc);
                    bufB[j * lp + t] = acc;
                }
            }
        }
        __syncthreads();
        // time difference (signature_algs.py:180) or a copy
        for (int ch = 0; ch < nchunk; ++ch) {
            const int t = ch * 64 + lane;
            if (t < l) {
                for (int j = wave; j < c; j += NW) {
                    const double f0 = bufB[j * lp + t];
                    U[j * lp + t] = A.difference ? bufB[j * lp + t + 1] - f0 : f0;
                }
            }
        }
        __syncthreads();
        // ---- level 1 and the exclusive running sums for level 2: thread = column
        if (threadIdx.x == 0) phi[0] = 1.0;
        for (int j = threadIdx.x; j < c; j += THREADS) {
            double run = 0.0;
            const double* u = U + size_t(j) * lp;
            double* e = bufA + size_t(j) * lp;
            const bool more = A.M >= 2;
#pragma unroll 8
            for (int t = 0; t < l; ++t) {
                const double v = u[t];
                if (more) e[t] = run;
                run += v;
            }
            phi[1 + j] = run;                                                                   // signature_algs.py:182
        }
        __syncthreads();
        double* cur = bufA;
        double* nxt = bufB;
        for (int lev = 2; lev <= A.M; ++lev) {
            const lr_const_ptr<int32_t> colptr = lr_as_const(A.sk[lev - 2].colptr);
            const lr_const_ptr<LrEntry> ent = lr_as_const(A.sk[lev - 2].ent);
            // P_lev[t][j] = sum_e val * U[t][i1] * E[t][i2]                                   low_rank_calculations.py:64-193
            for (int j = wave; j < r; j += NW) {
                const int e0 = colptr[j], e1 = colptr[j + 1];
                for (int ch = 0; ch < nchunk; ++ch) {
                    const int t = ch * 64 + lane;
                    const int tt = t < l ? t : 0;                 // idle lanes read a valid address
                    double acc = 0.0;
#pragma unroll UNROLL
                    for (int e = e0; e < e1; ++e) {
                        const double val = ent[e].val;            // (member by member: an address-space-4 struct has no copy constructor)
                        const int i1 = ent[e].i1, i2 = ent[e].i2;
                        acc = fma(val * U[i1 * lp + tt], cur[i2 * lp + tt], acc);
                    }
                    if (t < l) nxt[j * lp + t] = acc;
                }
            }
            __syncthreads();
            const int off = 1 + c + (lev - 2) * r;
            const bool more = lev < A.M;
            for (int j = threadIdx.x; j < r; j += THREADS) {
                double run = 0.0;
                double* e = nxt + size_t(j) * lp;
#pragma unroll 8
                for (int t = 0; t < l; ++t) {
                    const double v = e[t];
                    if (more) e[t] = run;                                                       // signature_algs.py:186
                    run += v;
                }
                phi[off + j] = run;                                                             // :191
            }
            __syncthreads();
            double* tmp = cur; cur = nxt; nxt = tmp;
        }
        // cur / nxt are read again by the next sequence's phase 0 (bufB) only after the barrier above
    }
}

// ---- the same with TWO arrays in LDS instead of three --------------------------------------------------------------------
// For sequences of at most 64 time steps and at most LR_FUSED2_COLS output columns per wavefront, a wavefront keeps the columns it
// produces in REGISTERS until every wavefront of the workgroup has finished reading the array they replace, and writes them in
// place after a barrier: no third buffer (the output of a sketch / of the whitening product), 52 KB instead of 78 KB of LDS at
// BASELINE configs[2]'s sequences with the reference's default ranks -- three workgroups per CU instead of two, i.e. half as many
// wavefronts again to hide the scalar-load latency of the sketch entries and the phases in which one wavefront works.
//   W holds: the scaled observations (transposed) -> kxs -> E_2 (running sums of U) -> P_2 -> E_3 -> ...;  U: feat -> U.
constexpr int LR_FUSED2_COLS = 8;
template <int THREADS, int UNROLL>
__global__ __launch_bounds__(THREADS) void lr_seq_features_fused2_kernel(LrFusedArgs A) {
    extern __shared__ double lr_lds[];
    const int lp = A.lp, c = A.c, r = A.r, L = A.L;
    double* const U = lr_lds;                               // [max(c, d_eff)][lp]
    double* const W = U + size_t(A.rows_b) * lp;            // [rows_b][lp]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NW = THREADS / 64;
    const int d_eff = A.P.d_eff();
    const int l = A.difference ? L - 1 : L;
    const int t = lane, tt = t < l ? t : 0;

    for (int64_t n = blockIdx.x; n < A.N; n += gridDim.x) {
        const double* Xn = A.X + n * int64_t(L) * A.P.d_in;
        double* phi = A.Phi + n * int64_t(A.F);
        // ---- scaled observations, U[fe][t]
        for (int q = threadIdx.x; q < L * d_eff; q += THREADS) {
            const int tq = q / d_eff, fe = q - tq * d_eff;
            U[fe * lp + tq] = scaled_point<double>(Xn, L, tq, fe, A.P);
        }
        __syncthreads();
        // ---- kxs, W[i][t]
        if (t < L) {
            double xs = 0.0;
            for (int fe = 0; fe < d_eff; ++fe) { const double x = U[fe * lp + t]; xs = fma(x, x, xs); }
            for (int i = wave; i < c; i += NW) {
                const lr_const_ptr<double> Si = lr_as_const(A.S) + size_t(i) * d_eff;
                double ip = 0.0, ss = 0.0;
                for (int fe = 0; fe < d_eff; ++fe) {
                    const double y = Si[fe];
                    ip = fma(U[fe * lp + t], y, ip);
                    ss = fma(y, y, ss);
                }
                W[i * lp + t] = base_eval<double>(A.kind, ip, xs, ss, A.p0, A.p1);
            }
        }
        __syncthreads();
        // ---- whitening into U (the observations are no longer needed): feat[j][t] = sum_i W[i][t] * Wh[i][j]
        {
            const lr_const_ptr<double> Wh = lr_as_const(A.Wh);
            const int tr = t < L ? t : 0;
#pragma unroll
            for (int k = 0; k < LR_FUSED2_COLS; ++k) {
                const int j = wave + k * NW;
                if (j < c) {
                    double acc = 0.0;
#pragma unroll 4
                    for (int i = 0; i < c; ++i) acc = fma(W[i * lp + tr], Wh[size_t(i) * c + j], acc);
                    if (t < L) U[j * lp + t] = acc;
                }
            }
        }
        __syncthreads();
        // time difference in place (signature_algs.py:180): a column belongs to one wavefront, whose lanes all read before any writes
#pragma unroll
        for (int k = 0; k < LR_FUSED2_COLS; ++k) {
            const int j = wave + k * NW;
            if (j < c && A.difference) {
                const double f0 = U[j * lp + tt], f1 = U[j * lp + tt + 1];
                if (t < l) U[j * lp + t] = f1 - f0;
            }
        }
        __syncthreads();
        // ---- level 1 and the exclusive running sums for level 2 (thread = column), W = E_2
        if (threadIdx.x == 0) phi[0] = 1.0;
        for (int j = threadIdx.x; j < c; j += THREADS) {
            double run = 0.0;
            const double* u = U + size_t(j) * lp;
            double* e = W + size_t(j) * lp;
            const bool more = A.M >= 2;
#pragma unroll 8
            for (int q = 0; q < l; ++q) {
                const double v = u[q];
                if (more) e[q] = run;
                run += v;
            }
            phi[1 + j] = run;                                                                   // signature_algs.py:182
        }
        __syncthreads();
        for (int lev = 2; lev <= A.M; ++lev) {
            const lr_const_ptr<int32_t> colptr = lr_as_const(A.sk[lev - 2].colptr);
            const lr_const_ptr<LrEntry> ent = lr_as_const(A.sk[lev - 2].ent);
            double out[LR_FUSED2_COLS];
#pragma unroll
            for (int k = 0; k < LR_FUSED2_COLS; ++k) {
                const int j = wave + k * NW;
                double acc = 0.0;
                if (j < r) {
                    const int e0 = colptr[j], e1 = colptr[j + 1];
#pragma unroll UNROLL
                    for (int e = e0; e < e1; ++e) {
                        const double val = ent[e].val;
                        const int i1 = ent[e].i1, i2 = ent[e].i2;
                        acc = fma(val * U[i1 * lp + tt], W[i2 * lp + tt], acc);
                    }
                }
                out[k] = acc;
            }
            __syncthreads();                                 // every wavefront has read E: P takes its place
#pragma unroll
            for (int k = 0; k < LR_FUSED2_COLS; ++k) {
                const int j = wave + k * NW;
                if (j < r && t < l) W[j * lp + t] = out[k];
            }
            __syncthreads();
            const int off = 1 + c + (lev - 2) * r;
            const bool more = lev < A.M;
            for (int j = threadIdx.x; j < r; j += THREADS) {
                double run = 0.0;
                double* e = W + size_t(j) * lp;
#pragma unroll 8
                for (int q = 0; q < l; ++q) {
                    const double v = e[q];
                    if (more) e[q] = run;                                                       // signature_algs.py:186
                    run += v;
                }
                phi[off + j] = run;                                                             // :191
            }
            __syncthreads();
        }
    }
}

// ---- inducing tensors ---------------------------------------------------------------------------------------------------
// One workgroup per tensor t.  Rows (k, e) of the tensor's lt * E components: scaled (kernels.py:367-398), kappa against the
// landmarks (low_rank_calculations.py:59), whitened (:60), differenced over e for incremental tensors (kernels.py:304); then
// level i is the chain  R = U[k];  R = sketch_{j-1}(U[k + j], R), j = 1 .. i-1  (signature_algs.py:211-221), thread = output
// column.  The sketches' entries come from L2 (they are the same for every tensor).  A few hundred multiply-adds per thread:
// the point is one launch instead of about twenty.
constexpr int LR_TENS_THREADS = 128;
__global__ __launch_bounds__(LR_TENS_THREADS) void lr_tens_features_fused_kernel(LrTensFusedArgs A) {
    extern __shared__ double lr_lds[];
    const int c = A.c, r = A.r, lt = A.lt, E = A.E, d_eff = A.P.d_eff();
    const int rows = lt * E, w = c > r ? c : r;
    double* const zs = lr_lds;                       // [rows][d_eff]
    double* const kx = zs + rows * d_eff;            // [rows][c]
    double* const ft = kx + rows * c;                // [rows][c]
    double* const U = ft + rows * c;                 // [lt][c]
    double* Ra = U + lt * c;                         // [w]
    double* Rb = Ra + w;                             // [w]
    const int64_t t = blockIdx.x;
    double* phi = A.Phi + t * int64_t(A.F);
    for (int q = threadIdx.x; q < rows * d_eff; q += LR_TENS_THREADS) {
        const int row = q / d_eff, fe = q - row * d_eff;
        const int k = row / E, e = row - k * E;
        const int lag = fe / A.P.d_in, f = fe - lag * A.P.d_in;
        double x = A.Z[((int64_t(k) * A.T + t) * E + e) * d_eff + fe];
        if (A.P.has_ls) {                            // kernels.py:374-379 / :391-395
            x = x / A.P.lsv(f);
            if (A.P.num_lags > 0) x = x * A.P.gamma[lag];
        }
        zs[q] = x;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < rows * c; q += LR_TENS_THREADS) {
        const int row = q / c, i = q - row * c;
        double ip = 0.0, xs = 0.0, ss = 0.0;
        for (int fe = 0; fe < d_eff; ++fe) {
            const double x = zs[row * d_eff + fe], y = A.S[size_t(i) * d_eff + fe];
            ip = fma(x, y, ip); xs = fma(x, x, xs); ss = fma(y, y, ss);
        }
        kx[q] = base_eval<double>(A.kind, ip, xs, ss, A.p0, A.p1);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < rows * c; q += LR_TENS_THREADS) {
        const int row = q / c, j = q - row * c;
        double acc = 0.0;
        for (int i = 0; i < c; ++i) acc = fma(kx[row * c + i], A.Wh[size_t(i) * c + j], acc);
        ft[q] = acc;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < lt * c; q += LR_TENS_THREADS) {
        const int k = q / c, j = q - k * c;
        U[q] = E == 2 ? ft[(k * 2 + 1) * c + j] - ft[(k * 2) * c + j] : ft[k * c + j];
    }
    __syncthreads();
    if (threadIdx.x == 0) phi[0] = 1.0;
    int k = 0;
    for (int i = 1; i <= A.M; ++i) {
        const double* R = U + k * c;
        int kw = c;
        ++k;
        double* cur = Ra;
        double* nxt = Rb;
        for (int j = 1; j < i; ++j) {
            const LrFusedSketch sk = A.sk[j - 1];
            const double* Uk = U + k * c;
            for (int jo = threadIdx.x; jo < r; jo += LR_TENS_THREADS) {
                double acc = 0.0;
                for (int e = sk.colptr[jo]; e < sk.colptr[jo + 1]; ++e) acc = fma(sk.ent[e].val * Uk[sk.ent[e].i1], R[sk.ent[e].i2], acc);
                cur[jo] = acc;
            }
            __syncthreads();
            R = cur;
            kw = r;
            double* tmp = cur; cur = nxt; nxt = tmp;
            ++k;
        }
        const int off = i == 1 ? 1 : 1 + c + (i - 2) * r;
        for (int jo = threadIdx.x; jo < kw; jo += LR_TENS_THREADS) phi[off + jo] = R[jo];
        __syncthreads();                             // R (Ra / Rb) is rewritten by the next level's chain
    }
}

}  // namespace gpsig
