// sig_feat_kernel.hpp -- SignatureLinear (first-order algorithm; higher orders: sig_horner below) as an inner product of explicit level features.
//
// For the LINEAR state-space kernel the increment lattice of a pair factorises, dM[a][b] = <dx_a, dy_b> (gpsig/kernels.py:799-806 into
// signature_algs.py:25-26), and with it the whole first-order recursion (signature_algs.py:28-35): level m is the sum over strictly
// increasing index tuples a_1 < .. < a_m, b_1 < .. < b_m of prod_i <dx_{a_i}, dy_{b_i}>, i.e.
//     K_m(x, y) = < Phi_m(x), Phi_m(y) >,     Phi_m(x) = sum_{a_1 < .. < a_m} dx_{a_1} (x) .. (x) dx_{a_m}   in (R^d)^{(x) m},
// d^m numbers per sequence and level, built by one sweep over time,  Phi_m <- Phi_m + Phi_{m-1}(previous step) (x) dx_a.
// A Gram entry then costs 2 sum_m d^m flops instead of the lattice sweep's L1 L2 (2d + 3M - 1): at BASELINE configs[1] (L = 64, d = 8,
// M = 5) 74.9 k against 119 k -- and the N x N batch of them is ONE contraction of depth 37,448, which is what the matrix cores are
// for (north_star: "MFMA only where it is a true contraction"; the float64 matrix rate of this chip equals its vector rate, but a GEMM
// runs near it while the lattice sweep, with its hand-overs and latencies, reaches half).  The planner (sig_features_plan in api.hip)
// takes this route where it is the cheaper one and the feature matrix fits; everything else -- every other base kernel, long state
// spaces, many levels -- stays on the lattice kernels.  Results agree with them to rounding (different summation order).
//
//   sig_features_kernel   one workgroup per sequence, the whole state in registers (d^M / threads top-level values per thread plus a
//                         private copy of their ancestors: no barrier, no LDS traffic in the sweep); then the level norms |Phi_m|^2 (= the level diagonal K_m(x, x)), and the features scaled
//                         by sqrt(sigma variances[m] / (|Phi_m|^2 + jitter)) (kernels.py:430-433, :471) so that the level sum,
//                         the normalisation and the weights are all inside the contraction.
//   sig_gram_kernel       C = A B^T on v_mfma_f64_16x16x4 in 128 x 128 tiles staged through LDS; symmetric products visit the upper
//                         tile triangle only; the depth is split over several workgroups per tile so that the launch fills the chip
//                         evenly (528 tiles on 512 workgroup slots would take two rounds), each writing its own partial sum.
//   sig_gram_reduce_kernel adds the partial sums in a fixed order (deterministic), mirrors, sets the exact diagonal, or packs the
//                         owned entries of a row block (multi-GPU, gpsig_kernel_K_symm_rows_compact).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "aux_kernels.hpp"
#include "sig_pieces.hpp"

namespace gpsig {

constexpr int sig_ipow(int d, int m) {           // d^m, saturating at 2^30 (the lookup asks about shapes far beyond what is built)
    long long v = 1;
    for (int i = 0; i < m; ++i) { v *= d; if (v > (1ll << 30)) return 1 << 30; }
    return int(v);
}
constexpr int sig_lower_total(int d, int M) { int s = 0; for (int m = 1; m < M; ++m) s += sig_ipow(d, m); return s; }
constexpr int sig_feature_count(int d, int M) { int s = 0; for (int m = 1; m <= M; ++m) s += sig_ipow(d, m); return s; }
constexpr int SIG_MAX_TOP = 32768;               // d^M values of the top level: 64 per thread of a 512-thread workgroup (two wavefronts per SIMD,
                                                 // 256 registers; 1024 threads would have 128 and spill, fewer threads more than 64 values each)
constexpr int sig_threads(int d, int M) {
    int t = 64;
    while (t < 512 && t * 64 < sig_ipow(d, M)) t *= 2;
    return t;
}

struct SigFeatArgs {
    const double* X;        // (N, L, d_in) as the caller gives it
    int64_t N;
    int L, difference;
    ScaleParams P;
    const double* w;        // (M+1) sigma * variances (device) or NULL = 1
    int normalize;          // divide level m by sqrt(|Phi_m|^2 + jitter)
    double jitter;
    double* Phi;            // (N, ld): levels 1..M, then the level-0 column, then zeros up to ld
    int64_t ld;
    double* dlev;           // (N, M+1) raw level diagonals |Phi_m|^2 (level 0: 1), or NULL
    int order;              // 1: signature_algs.py:8-35; > 1: the higher-order algorithm (:37-74), see sig_horner below
    int unit_points;        // SignatureCosine (kernels.py:820-828): <x, y> / (|x| |y|) is the linear kernel of the points x / |x|
    int norm_squared;       // this side is divided by (|Phi_m|^2 + jitter), not by its square root (K_seq_n_seq_covs: kernels.py:713 + :750)
    int natural_order;      // every level in the natural order of its multi-indices (last index fastest): not the sibling kernel's own order
};

// Higher orders (signature_algs.py:37-74: a step may repeat an index up to `order` times, with 1 / k! for k repeats).  For the linear
// base kernel that algorithm's level m is the inner product of the features obtained by multiplying, step by step, with the exponential
// of the increment truncated at degree `order` (Chen; order = num_levels: the signature of the piecewise-linear path, what the
// reference's notebook checks against esig):  Phi_m <- sum_{k = 0 .. min(order, m)} Phi_{m-k} (x) dx^(x)k / k!.  Along one chain of
// entries -- V[m] the level-m entry, E[m] the component of dx that extends level m-1 to it -- the sum is a Horner form,
//     new V[m] = V[m] + E[m] H_1,   H_j = V[m-j] + E[m-j] / (j+1) H_{j+1},   H_{min(order, m)} = V[m - min(order, m)],
// and sig_horner returns H_jmin (1: the whole bracket; 2: what is left when the level below is folded in by the caller).  V[0] = 1,
// E[0] = 0.  About d^(m-1) extra multiply-adds per level and step: the higher orders cost the features a fifth more, the contraction
// nothing -- against pair kernels that take 34 to 150 times the first order's time at BASELINE configs[1]'s size.
template <int MM>
__device__ __forceinline__ double sig_horner(const double (&V)[MM], const double (&E)[MM], int m, int order, int jmin) {
    double h = 0.0;
#pragma unroll
    for (int j = MM; j >= 1; --j)
        if (j <= m && j >= jmin && j <= order) h = fma(h * E[m - j], 1.0 / (j + 1), V[m - j]);
    return h;
}

// D = the number of columns after lags (not padded: the feature count is D^m), M = num_levels >= 2.
// Everything a thread needs lives in its registers: thread t owns the entries t, t + T, .. of level M-1 (its "parents"), their D
// children each of level M, and a private copy of every ANCESTOR of each parent (one entry per lower level -- M - 2 extra multiply-adds
// per parent and step, recomputed by every thread that shares the ancestor, against D + 1 useful ones).  So the sweep over time needs no
// barrier and no LDS traffic besides the broadcast reads of the increment: the first form kept the lower levels in LDS behind a barrier
// per step and ran at a third of its issue rate on the LDS round trips (1.16 ms for BASELINE configs[1]'s 4,096 sequences; this: see DESIGN).
template <int D, int M, bool HO>
__global__ __launch_bounds__(sig_threads(D, M)) void sig_features_kernel(const SigFeatArgs A) {
    constexpr int T = sig_threads(D, M);
    constexpr int NTOP = sig_ipow(D, M), NPAR = sig_ipow(D, M - 1);     // top-level values, their parents (level M-1)
    constexpr int PPT = (NPAR + T - 1) / T;                             // parents per thread
    constexpr int NA = M - 1;                                           // ancestors kept per parent: levels M-1 (k = 0) .. 1 (k = M-2)
    constexpr int S = D <= 64 ? 64 / D : 1;                             // steps whose increments one wavefront register holds
    static_assert(D <= 64, "one row of increments per wavefront register");
    extern __shared__ double sf_sm[];
    double* const dx = sf_sm;                       // R x D increments of this sequence, then zeros up to (L + S) x D + 64
    double* const red = dx + (size_t(A.L) + S) * D + 64;       // T / 64 partial sums
    __shared__ double norms[M + 1];
    const int tid = threadIdx.x;
    const int R = A.difference ? A.L - 1 : A.L;
    // Which component of dx extends the ancestor at distance k of parent q (that ancestor's last index, (q T + tid) / D^k mod D):
    // the same for every q where D^(k+1) divides T (one register per level), a compile-time constant where T divides D^k (the
    // register of d_ itself), one register per (q, k) otherwise (small D^M only).
    int comp_k[NA], comp_g[PPT][NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        comp_k[k] = (tid / sig_ipow(D, k)) % D;
#pragma unroll
        for (int q = 0; q < PPT; ++q) comp_g[q][k] = ((q * T + tid) / sig_ipow(D, k)) % D;
    }
    auto opaque = [](int i) { asm volatile("" : "+v"(i)); return i; };
    // the thread that reports a shared ancestor: the first of the threads holding it
    auto owner = [&](int q, int k) { const int idx = q * T + tid; return idx < NPAR && idx % sig_ipow(D, k) == 0; };
    for (int64_t n = blockIdx.x; n < A.N; n += gridDim.x) {
        const double* Xn = A.X + n * int64_t(A.L) * A.P.d_in;
        __syncthreads();                            // the previous sequence's increments are no longer read
        if (!A.unit_points) {
            for (int e = tid; e < (A.L + S) * D + 64; e += T) {
                const int a = e / D, f = e - a * D;
                dx[e] = a >= R ? 0.0 : A.difference ? scaled_point<double>(Xn, A.L, a + 1, f, A.P) - scaled_point<double>(Xn, A.L, a, f, A.P)
                                     : scaled_point<double>(Xn, A.L, a, f, A.P);
            }
        } else {                                    // the scaled points, their norms, then the (increments of the) unit vectors
            double* const pts = red + 64;           // L x D, then L reciprocal norms
            double* const inv = pts + size_t(A.L) * D;
            for (int e = tid; e < A.L * D; e += T) pts[e] = scaled_point<double>(Xn, A.L, e / D, e % D, A.P);
            __syncthreads();
            for (int a = tid; a < A.L; a += T) {
                double ss = 0.0;
                for (int f = 0; f < D; ++f) ss = fma(pts[a * D + f], pts[a * D + f], ss);
                inv[a] = 1.0 / sqrt(ss);
            }
            __syncthreads();
            for (int e = tid; e < (A.L + S) * D + 64; e += T) {
                const int a = e / D, f = e - a * D;
                dx[e] = a >= R ? 0.0 : A.difference ? pts[(a + 1) * D + f] * inv[a + 1] - pts[a * D + f] * inv[a] : pts[a * D + f] * inv[a];
            }
        }
        double top[PPT][D], anc[PPT][NA];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
#pragma unroll
            for (int k = 0; k < NA; ++k) anc[q][k] = 0.0;
#pragma unroll
            for (int f = 0; f < D; ++f) top[q][f] = 0.0;
        }
        __syncthreads();
        // The increment of a step is the same for every thread: it is read from LDS once per S steps -- lane l of a wavefront holds
        // element l of the S rows dx[a0 .. a0 + S) -- and handed to the multiply-adds as a SCALAR operand (v_readlane), so a step waits
        // for no LDS round trip of its own.  Rows past the last increment are zeros (they change nothing).
        for (int a0 = 0; a0 < R; a0 += S) {
            const double rows = dx[a0 * D + (tid & 63)];
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const double* dxa = dx + (a0 + i) * D;
                double d_[D];
#pragma unroll
                for (int f = 0; f < D; ++f) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(rows), i * D + f);
                    const int hi = __builtin_amdgcn_readlane(__double2hiint(rows), i * D + f);
                    d_[f] = __hiloint2double(hi, lo);
                }
                // the ancestors' components of dx differ by lane: from LDS by index (behind an opaque index: a select among the D
                // values of d_, which the compiler builds when it can see the index is one of them, costs 2 D instructions)
                double dck[NA];
#pragma unroll
                for (int k = 0; k < NA; ++k) dck[k] = dxa[opaque(comp_k[k])];
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
                    double ec[NA];                  // the component of dx that extends the level below into ancestor k
#pragma unroll
                    for (int k = 0; k < NA; ++k) {
                        if (T % sig_ipow(D, k + 1) == 0) ec[k] = dck[k];
                        else if (sig_ipow(D, k) % T == 0) ec[k] = d_[(q / (sig_ipow(D, k) / T > 0 ? sig_ipow(D, k) / T : 1)) % D];
                        else ec[k] = dxa[opaque(comp_g[q][k])];
                    }
                    if constexpr (!HO) {
                        // every level from the OLD value of the level below it: the top first, then the ancestors from the highest down
#pragma unroll
                        for (int f = 0; f < D; ++f) top[q][f] = fma(anc[q][0], d_[f], top[q][f]);
#pragma unroll
                        for (int k = 0; k < NA; ++k) anc[q][k] = fma(k + 1 < NA ? anc[q][k + 1] : 1.0, ec[k], anc[q][k]);
                    } else {
                        // higher orders: the chain of this parent by level (sig_horner above), old values throughout
                        double V[M], E[M];
                        V[0] = 1.0; E[0] = 0.0;
#pragma unroll
                        for (int m = 1; m < M; ++m) { V[m] = anc[q][M - 1 - m]; E[m] = ec[M - 1 - m]; }
                        const double ht = sig_horner<M>(V, E, M, A.order, 1);
#pragma unroll
                        for (int f = 0; f < D; ++f) top[q][f] = fma(ht, d_[f], top[q][f]);
#pragma unroll
                        for (int k = 0; k < NA; ++k) anc[q][k] = fma(sig_horner<M>(V, E, M - 1 - k, A.order, 1), ec[k], anc[q][k]);
                    }
                }
            }
        }
        // level norms: |Phi_m|^2 = K_m(x, x); an ancestor shared by several threads is counted by its owner
        for (int m = 1; m <= M; ++m) {
            double s = 0.0;
            if (m == M) {
#pragma unroll
                for (int q = 0; q < PPT; ++q)
                    if (owner(q, 0))
#pragma unroll
                        for (int f = 0; f < D; ++f) s = fma(top[q][f], top[q][f], s);
            } else {
                const int k = M - 1 - m;
#pragma unroll
                for (int q = 0; q < PPT; ++q)
#pragma unroll
                    for (int kk = 0; kk < NA; ++kk)
                        if (kk == k && owner(q, kk)) s = fma(anc[q][kk], anc[q][kk], s);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if ((tid & 63) == 0) red[tid >> 6] = s;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                for (int k2 = 0; k2 < (T + 63) / 64; ++k2) t += red[k2];
                norms[m] = t;
            }
            __syncthreads();
        }
        double* out = A.Phi + n * A.ld;
        auto scale_of = [&](int m) {
            const double w = A.w ? A.w[m] : 1.0;
            const double nm = m == 0 ? 1.0 : norms[m];
            return A.normalize ? (A.norm_squared ? sqrt(w) / (nm + A.jitter) : sqrt(w / (nm + A.jitter))) : sqrt(w);
        };
        // levels 1 .. M-1 from their owners, level M from everybody
        {
            int off = 0;
#pragma unroll
            for (int m = 1; m < M; ++m) {
                const int k = M - 1 - m;
                const double sc = scale_of(m);
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
#pragma unroll
                    for (int kk = 0; kk < NA; ++kk)
                        if (kk == k && owner(q, kk)) out[off + (q * T + tid) / sig_ipow(D, kk)] = sc * anc[q][kk];
                }
                off += sig_ipow(D, m);
            }
            const double sc = scale_of(M);
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const int pi = q * T + tid;
                if (pi < NPAR) {
                    double* o = out + off + pi * D;             // D consecutive doubles per lane, consecutive lanes side by side
                    if constexpr (D % 2 == 0) {                 // 16-byte stores (even D: the level offsets and the row stride are even)
                        double2* o2 = reinterpret_cast<double2*>(o);
#pragma unroll
                        for (int f = 0; f < D; f += 2) o2[f / 2] = double2{sc * top[q][f], sc * top[q][f + 1]};
                    } else {
#pragma unroll
                        for (int f = 0; f < D; ++f) o[f] = sc * top[q][f];
                    }
                }
            }
            const int F = off + NTOP;
            for (int64_t e = F + tid; e < A.ld; e += T) out[e] = e == F ? scale_of(0) : 0.0;      // level 0 == 1 (signature_algs.py:20)
        }
        if (A.dlev && tid <= M) A.dlev[n * (M + 1) + tid] = tid == 0 ? 1.0 : norms[tid];
    }
}

// The same features for shapes with D^(M-2) threads (d = 8: M = 4, 5 -- the headline): a thread's D parents are SIBLINGS (level M-1
// entries t D .. t D + D-1), so it holds ONE entry of level M-2 (entry t: nobody else does), one of each lower level (shared with the
// D^(k-1) threads below the same entry) and no redundant copies of level M-1: D^2 + D + (M-2) multiply-adds per step instead of
// D (D + M - 1), 75 instead of 96 at d = 8, M = 5, and a parent's component of the increment is a compile-time lane of the row
// register.  Within a level the features are stored in whatever order makes the stores whole lines (the contraction only needs both
// of its operands in the same order, and levels in their places): level M as (child pair, thread), level M-1 as (sibling, thread).
constexpr bool sig_siblings(int d, int M) { return M >= 3 && sig_threads(d, M) == sig_ipow(d, M - 2); }

template <int D, int M, bool HO>
__global__ __launch_bounds__(sig_threads(D, M)) void sig_features_sib_kernel(const SigFeatArgs A) {
    constexpr int T = sig_threads(D, M);
    static_assert(T == sig_ipow(D, M - 2) && M >= 3 && D <= 64, "one level M-2 entry per thread");
    constexpr int NW = (T + 63) / 64;
    constexpr int NANC = M - 2;                                         // a[k], k = 1 .. M-2: level M-1-k, entry t / D^(k-1)
    constexpr int S = 64 / D;
    extern __shared__ double sf_sm[];
    double* const dx = sf_sm;                       // R x D increments of this sequence, then zeros up to (L + S) x D + 64
    double* const red = dx + (size_t(A.L) + S) * D + 64;       // NW x (M + 1) partial sums
    __shared__ double norms[M + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = A.difference ? A.L - 1 : A.L;
    int comp[NANC + 1];
    bool own[NANC + 1];
#pragma unroll
    for (int k = 1; k <= NANC; ++k) {
        comp[k] = (tid / sig_ipow(D, k - 1)) % D;
        own[k] = tid % sig_ipow(D, k - 1) == 0;
    }
    auto opaque = [](int i) { asm volatile("" : "+v"(i)); return i; };
    for (int64_t n = blockIdx.x; n < A.N; n += gridDim.x) {
        const double* Xn = A.X + n * int64_t(A.L) * A.P.d_in;
        __syncthreads();                            // the previous sequence's increments are no longer read
        if (!A.unit_points) {
            for (int e = tid; e < (A.L + S) * D + 64; e += T) {
                const int a = e / D, f = e - a * D;
                dx[e] = a >= R ? 0.0 : A.difference ? scaled_point<double>(Xn, A.L, a + 1, f, A.P) - scaled_point<double>(Xn, A.L, a, f, A.P)
                                     : scaled_point<double>(Xn, A.L, a, f, A.P);
            }
        } else {                                    // the scaled points, their norms, then the (increments of the) unit vectors
            double* const pts = red + 64;           // L x D, then L reciprocal norms
            double* const inv = pts + size_t(A.L) * D;
            for (int e = tid; e < A.L * D; e += T) pts[e] = scaled_point<double>(Xn, A.L, e / D, e % D, A.P);
            __syncthreads();
            for (int a = tid; a < A.L; a += T) {
                double ss = 0.0;
                for (int f = 0; f < D; ++f) ss = fma(pts[a * D + f], pts[a * D + f], ss);
                inv[a] = 1.0 / sqrt(ss);
            }
            __syncthreads();
            for (int e = tid; e < (A.L + S) * D + 64; e += T) {
                const int a = e / D, f = e - a * D;
                dx[e] = a >= R ? 0.0 : A.difference ? pts[(a + 1) * D + f] * inv[a + 1] - pts[a * D + f] * inv[a] : pts[a * D + f] * inv[a];
            }
        }
        double top[D][D], par[D], anc[NANC + 1];
#pragma unroll
        for (int q = 0; q < D; ++q) {
            par[q] = 0.0;
#pragma unroll
            for (int f = 0; f < D; ++f) top[q][f] = 0.0;
        }
#pragma unroll
        for (int k = 1; k <= NANC; ++k) anc[k] = 0.0;
        __syncthreads();
        for (int a0 = 0; a0 < R; a0 += S) {
            const double rows = dx[a0 * D + lane];                  // S rows of increments per wavefront register (zeros past the last)
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const double* dxa = dx + (a0 + i) * D;
                double d_[D], dc[NANC + 1];
#pragma unroll
                for (int f = 0; f < D; ++f) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(rows), i * D + f);
                    const int hi = __builtin_amdgcn_readlane(__double2hiint(rows), i * D + f);
                    d_[f] = __hiloint2double(hi, lo);
                }
#pragma unroll
                for (int k = 1; k <= NANC; ++k) dc[k] = dxa[opaque(comp[k])];
                if constexpr (!HO) {
                    // every level from the OLD value of the level below it: top, parents, then the ancestors from the highest down
#pragma unroll
                    for (int q = 0; q < D; ++q)
#pragma unroll
                        for (int f = 0; f < D; ++f) top[q][f] = fma(par[q], d_[f], top[q][f]);
#pragma unroll
                    for (int q = 0; q < D; ++q) par[q] = fma(anc[1], d_[q], par[q]);
#pragma unroll
                    for (int k = 1; k < NANC; ++k) anc[k] = fma(anc[k + 1], dc[k], anc[k]);
                    anc[NANC] += dc[NANC];
                } else {
                    // higher orders (sig_horner above): the thread's chain below the parents is the same for all of them
                    double V[M], E[M];
                    V[0] = 1.0; E[0] = 0.0; V[M - 1] = 0.0; E[M - 1] = 0.0;        // (level M-1: the parents, folded in below)
#pragma unroll
                    for (int m = 1; m <= M - 2; ++m) { V[m] = anc[M - 1 - m]; E[m] = dc[M - 1 - m]; }
                    const double g2 = 0.5 * sig_horner<M>(V, E, M, A.order, 2);     // the bracket behind a parent's own term, over 2
#pragma unroll
                    for (int q = 0; q < D; ++q) {
                        const double hq = fma(g2, d_[q], par[q]);
#pragma unroll
                        for (int f = 0; f < D; ++f) top[q][f] = fma(hq, d_[f], top[q][f]);
                    }
                    const double hp = sig_horner<M>(V, E, M - 1, A.order, 1);
#pragma unroll
                    for (int q = 0; q < D; ++q) par[q] = fma(hp, d_[q], par[q]);
#pragma unroll
                    for (int k = 1; k <= NANC; ++k) anc[k] = fma(sig_horner<M>(V, E, M - 1 - k, A.order, 1), dc[k], anc[k]);
                }
            }
        }
        // level norms |Phi_m|^2 (= K_m(x, x)), all levels in one pass
        {
            double sq[M + 1];
#pragma unroll
            for (int m = 1; m <= M; ++m) sq[m] = 0.0;
#pragma unroll
            for (int q = 0; q < D; ++q) {
                sq[M - 1] = fma(par[q], par[q], sq[M - 1]);
#pragma unroll
                for (int f = 0; f < D; ++f) sq[M] = fma(top[q][f], top[q][f], sq[M]);
            }
#pragma unroll
            for (int k = 1; k <= NANC; ++k)
                if (own[k]) sq[M - 1 - k] = anc[k] * anc[k];
#pragma unroll
            for (int m = 1; m <= M; ++m) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sq[m] += __shfl_xor(sq[m], o, 64);
                if (lane == 0) red[wave * (M + 1) + m] = sq[m];
            }
            __syncthreads();
            if (tid >= 1 && tid <= M) {
                double t = 0.0;
                for (int k2 = 0; k2 < NW; ++k2) t += red[k2 * (M + 1) + tid];
                norms[tid] = t;
            }
            __syncthreads();
        }
        double* out = A.Phi + n * A.ld;
        auto scale_of = [&](int m) {
            const double w = A.w ? A.w[m] : 1.0;
            const double nm = m == 0 ? 1.0 : norms[m];
            return A.normalize ? (A.norm_squared ? sqrt(w) / (nm + A.jitter) : sqrt(w / (nm + A.jitter))) : sqrt(w);
        };
        int off = 0;
#pragma unroll
        for (int m = 1; m <= M - 2; ++m) {                       // levels 1 .. M-2 from the ancestors' owners
            const int k = M - 1 - m;
            if (own[k]) out[off + tid / sig_ipow(D, k - 1)] = scale_of(m) * anc[k];
            off += sig_ipow(D, m);
        }
        {
            const double sc = scale_of(M - 1);                   // level M-1: (sibling, thread)
#pragma unroll
            for (int q = 0; q < D; ++q) out[off + q * T + tid] = sc * par[q];
            off += sig_ipow(D, M - 1);
        }
        {
            const double sc = scale_of(M);                       // level M: (parent, child pair, thread), 16 bytes per lane
            if constexpr (D % 2 == 0) {
                double2* o2 = reinterpret_cast<double2*>(out + off);
#pragma unroll
                for (int q = 0; q < D; ++q)
#pragma unroll
                    for (int f = 0; f < D; f += 2) o2[(q * (D / 2) + f / 2) * T + tid] = double2{sc * top[q][f], sc * top[q][f + 1]};
            } else {
#pragma unroll
                for (int q = 0; q < D; ++q)
#pragma unroll
                    for (int f = 0; f < D; ++f) out[off + (q * D + f) * T + tid] = sc * top[q][f];
            }
            off += sig_ipow(D, M);
        }
        for (int64_t e = off + tid; e < A.ld; e += T) out[e] = e == off ? scale_of(0) : 0.0;      // level 0 == 1 (signature_algs.py:20)
        if (A.dlev && tid <= M) A.dlev[n * (M + 1) + tid] = tid == 0 ? 1.0 : norms[tid];
    }
}

inline size_t sig_features_lds_bytes(int d, int M, int L) {
    (void)M;
    return sizeof(double) * ((size_t(L) + (d <= 64 ? 64 / d : 1)) * d + 64 + 64 + size_t(L) * d + size_t(L));      // (+ points and norms: SignatureCosine)
}

// float32 calls: the contraction is a float64 computation (float64 matrix cores; a float32 accumulation over 37,000 products would not
// hold 1e-4 anyway) -- the sequences are widened on the way in, the result rounded on the way out.
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void sig_widen_kernel(const float* __restrict__ in, double* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) out[i] = double(in[i]);
}
#else
__global__ void sig_widen_kernel(const float* __restrict__ in, double* __restrict__ out, int64_t n);
#endif
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void sig_narrow_kernel(const double* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) out[i] = float(in[i]);
}
#else
__global__ void sig_narrow_kernel(const double* __restrict__ in, float* __restrict__ out, int64_t n);
#endif

// ---- C = A B^T, float64 matrix cores, depth split over workgroups ----------------------------------------------------------------
typedef double sig_f64x4 __attribute__((ext_vector_type(4)));
constexpr int SG_BM = 128, SG_BN = 128, SG_BK = 16, SG_LDK = SG_BK + 1;
constexpr int SG_MAX_PIECES = 144;        // depth pieces per tile (api.hip's planner: at most 128 equal ones, or a few more of graded sizes)

struct SigGramArgs {
    const double* A; const double* B;     // (NA, lda), (NB, ldb) row-major; B row of output column c is (b_off + c) mod b_mod
    int64_t NA, NB, lda, ldb;
    int64_t b_off, b_mod;
    int k_begin, k_end;                   // depth range of this product
    int nsplit;                           // workgroups per tile along the depth
    int bound[SG_MAX_PIECES + 1];         // piece s covers the slabs [bound[s], bound[s + 1]) of the depth range (sig_piece_bounds)
    int symmetric;                        // A == B, NA == NB, b_off == 0: tiles (bi <= bj) only
    int ntj;                              // tile columns
    double* part;                         // (nsplit, NA, NB) partial sums
    int64_t band;                         // > 0 (row blocks of a symmetric Gram): row i owns columns i .. i + band only -- tiles outside are skipped
};

// Which depth piece and which tile this workgroup computes.
__device__ inline void sig_tile_of(const SigGramArgs& G, int& split_out, int& bi, int& bj) {
    const int ntiles = gridDim.x / G.nsplit;
    const int split = blockIdx.x / ntiles;
    int tile;
    {
        // Workgroups go to the 8 XCDs round robin by their index, and each XCD has an L2 of its own: give XCD x a CONTIGUOUS range of
        // this split's tiles (in the block-major order below), so that the ~64 tiles it works on at a time share operand panels in its
        // L2 -- an 8 x 8 block of tiles reads 16 panels, 64 tiles strided over the whole triangle read all of them.
        const int j = blockIdx.x - split * ntiles, off = (split * ntiles) & 7, x = (off + j) & 7;
        int start = 0;
        for (int xp = 0; xp < x; ++xp) {
            const int j0 = (xp - off + 8) & 7;
            start += j0 < ntiles ? (ntiles - 1 - j0) / 8 + 1 : 0;
        }
        tile = start + ((j - ((x - off + 8) & 7)) >> 3);
    }
    {
        // tile -> (bi, bj): 8 x 8 blocks of tiles, block rows first; symmetric products keep the blocks and tiles with bi <= bj
        constexpr int SB = 8;
        const int nti = G.symmetric ? G.ntj : (ntiles / G.ntj), nbi = (nti + SB - 1) / SB, nbj = (G.ntj + SB - 1) / SB;
        int Bi = 0, Bj = 0, hi = 0, wj = 0;
        bool found = false;
        for (Bi = 0; Bi < nbi && !found; ++Bi) {
            hi = nti - Bi * SB < SB ? nti - Bi * SB : SB;
            for (Bj = G.symmetric ? Bi : 0; Bj < nbj; ++Bj) {
                wj = G.ntj - Bj * SB < SB ? G.ntj - Bj * SB : SB;
                const int cnt = (G.symmetric && Bj == Bi) ? hi * (hi + 1) / 2 : hi * wj;
                if (tile < cnt) { found = true; break; }
                tile -= cnt;
            }
            if (found) break;
        }
        int li_, lj_;
        if (G.symmetric && Bj == Bi) {            // the upper triangle of a diagonal block, row by row
            li_ = 0;
            int rowlen = hi;
            while (tile >= rowlen) { tile -= rowlen; ++li_; --rowlen; }
            lj_ = li_ + tile;
        } else {
            li_ = tile / wj;
            lj_ = tile - li_ * wj;
        }
        bi = Bi * SB + li_;
        bj = Bj * SB + lj_;
    }
    split_out = split;
}

// grid: nsplit * ntiles workgroups, split-major (the tiles of one depth chunk run together: they share operand slabs in L2).
// VEC: the depth range starts at an even column (16-byte loads of the operands); otherwise element by element.
template <bool VEC>
static __global__ __launch_bounds__(256, 2) void sig_gram_kernel(const SigGramArgs G) {
    __shared__ double As[2][SG_BM * SG_LDK];            // two slabs: the next one is written while this one is multiplied (one barrier per slab)
    __shared__ double Bs[2][SG_BN * SG_LDK];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    int split, bi, bj;
    sig_tile_of(G, split, bi, bj);
    const int64_t tile_i = int64_t(bi) * SG_BM, tile_j = int64_t(bj) * SG_BN;
    if (G.band > 0 && (tile_j + SG_BN - 1 < tile_i || tile_j > tile_i + SG_BM - 1 + G.band)) return;      // nothing of this tile is owned
    // depth chunk of this workgroup, in whole slabs
    const int s0 = G.bound[split], s1 = G.bound[split + 1];
    const int kb = G.k_begin + s0 * SG_BK, ke = (G.k_begin + s1 * SG_BK < G.k_end) ? G.k_begin + s1 * SG_BK : G.k_end;
    // staging: thread t fetches columns [8 h, 8 h + 8) of row t >> 1 of the slab, h = t & 1
    const int srow = tid >> 1, scol = (tid & 1) * 8;
    const int64_t ai = tile_i + srow, bjr = tile_j + srow;
    const bool aok = ai < G.NA, bok = bjr < G.NB;
    int64_t brow_i = G.b_off + (bok ? bjr : 0);
    if (brow_i >= G.b_mod) brow_i -= G.b_mod;
    const double* arow = G.A + (aok ? ai : 0) * G.lda;
    const double* brow = G.B + brow_i * G.ldb;
    double pa[8], pb[8];
    auto fetch = [&](int k0) {
        if (VEC && k0 + SG_BK <= ke) {
            const double2* a2 = reinterpret_cast<const double2*>(arow + k0 + scol);
            const double2* b2 = reinterpret_cast<const double2*>(brow + k0 + scol);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // rows past the end read row 0 instead (arow / brow above): what they produce is never stored, and a select here
                // would sit right behind the loads and make the wave wait for them before its multiplies instead of after
                const double2 va = a2[e], vb = b2[e];
                pa[2 * e] = va.x; pa[2 * e + 1] = va.y;
                pb[2 * e] = vb.x; pb[2 * e + 1] = vb.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + scol + e;
                pa[e] = (aok && k < ke) ? arow[k] : 0.0;
                pb[e] = (bok && k < ke) ? brow[k] : 0.0;
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            As[buf][srow * SG_LDK + scol + e] = pa[e];
            Bs[buf][srow * SG_LDK + scol + e] = pb[e];
        }
    };
    sig_f64x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = sig_f64x4{0.0, 0.0, 0.0, 0.0};
    if (kb < ke) {
        fetch(kb);
        stash(0);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = kb; k0 < ke; k0 += SG_BK, buf ^= 1) {
        const bool more = k0 + SG_BK < ke;
        if (more) fetch(k0 + SG_BK);                        // in flight while this slab is multiplied
#pragma unroll
        for (int kk = 0; kk < SG_BK; kk += 4) {
            double av[4], bv[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                av[m] = As[buf][(wr * 64 + m * 16 + li) * SG_LDK + kk + lk];
                bv[m] = Bs[buf][(wc * 64 + m * 16 + li) * SG_LDK + kk + lk];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[n], acc[m][n], 0, 0, 0);
            // the next slab goes to the other buffer (last read one slab ago, before the barrier that ended it) ahead of the last
            // quarter of this slab's multiplies: the loads have had three quarters of a slab to land, and the LDS writes finish
            // under the multiplies instead of in front of the barrier
            if (kk == SG_BK - 8 && more) stash(buf ^ 1);
        }
        __syncthreads();
    }
    double* const P = G.part + int64_t(split) * G.NA * G.NB;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = tile_i + wr * 64 + m * 16 + lk + 4 * r, j = tile_j + wc * 64 + n * 16 + li;
                if (i < G.NA && j < G.NB) P[i * G.NB + j] = acc[m][n][r];
            }
}


// The same product with the slabs brought in by LDS-DMA and the operand fragments prefetched across the barrier (whole slabs only:
// k_begin a multiple of SG_BK, rows padded with zeros up to a multiple of SG_BK, 16-byte aligned).  A slab is 128 rows x 16 doubles per
// operand, rows unpadded (the DMA writes lane-linear: 8 lanes x 16 bytes = one row) with the 16-byte slots of row r XORed by (r >> 1) & 7
// ON THE GLOBAL SIDE -- lane (row, slot s) fetches columns 2 (s ^ swz) -- so that a fragment read (16 rows x 4 depth, 8 bytes per lane)
// touches every bank pair once.  Per slab and wave: 8 DMA instructions at the top (no staging registers, no ds_write), fragments of
// depth step k+1 read while step k multiplies, ONE barrier in front of the last step's multiplies -- by then the wave holds that
// step's fragments, and the next slab's first fragments are read right behind the barrier, under 16 MFMAs.  Same summation order as
// sig_gram_kernel: bit-identical results.
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ __launch_bounds__(256, 2) void sig_gram_dma_kernel(const SigGramArgs G) {
    // (1 KiB unused in front: an LDS-DMA load adds its immediate offset to the LDS address as well as to the global one, so the
    // destinations below are given minus that offset -- down to 1 KiB below the first buffer)
    constexpr int SG_GUARD = 8 * SG_BK;
    __shared__ __attribute__((aligned(16))) double lds_all[SG_GUARD + 2 * SG_BM * SG_BK + 2 * SG_BN * SG_BK];
    double (*const As)[SG_BM * SG_BK] = reinterpret_cast<double (*)[SG_BM * SG_BK]>(lds_all + SG_GUARD);
    double (*const Bs)[SG_BN * SG_BK] = reinterpret_cast<double (*)[SG_BN * SG_BK]>(lds_all + SG_GUARD + 2 * SG_BM * SG_BK);
    // the wavefront's index as a SCALAR: everything that depends on it and on the buffer alone (the LDS-DMA destinations) stays in
    // the scalar unit -- beside the multiplies every vector instruction costs: 38 of them per slab and wave (pointer increments, LDS
    // address arithmetic, readfirstlanes) left the matrix pipes idle 7 % of a launch that never waited for memory
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    int split, bi, bj;
    sig_tile_of(G, split, bi, bj);
    const int64_t tile_i = int64_t(bi) * SG_BM, tile_j = int64_t(bj) * SG_BN;
    if (G.band > 0 && (tile_j + SG_BN - 1 < tile_i || tile_j > tile_i + SG_BM - 1 + G.band)) return;      // nothing of this tile is owned
    const int s0 = G.bound[split], s1 = G.bound[split + 1];
    const int nsl = s1 - s0;
    const double* ga[4];          // this lane's 16 bytes of the CURRENT slab, piece q; the next slab is 128 bytes on
    const double* gb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                   // piece q of this wave: rows (4 q + wave) 8 .. + 8, lane = (row, slot)
        const int r = (q * 4 + wave) * 8 + (lane >> 3), p = (lane & 7) ^ ((r >> 1) & 7);
        const int64_t ai = tile_i + r, bjr = tile_j + r;
        int64_t brow_i = G.b_off + (bjr < G.NB ? bjr : 0);
        if (brow_i >= G.b_mod) brow_i -= G.b_mod;
        ga[q] = G.A + (ai < G.NA ? ai : 0) * G.lda + G.k_begin + int64_t(s0) * SG_BK + 2 * p;
        gb[q] = G.B + brow_i * G.ldb + G.k_begin + int64_t(s0) * SG_BK + 2 * p;
    }
    // slab (current + AHEAD) into buffer BUF; AHEAD is an immediate offset of the load, so the pointers move once per 8 slabs
    auto dma = [&](auto buf_c, auto ahead_c) {
        constexpr int BUF = decltype(buf_c)::value, OFF = decltype(ahead_c)::value * SG_BK * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga[q],
                                             (__attribute__((address_space(3))) void*)(&As[BUF][(q * 4 + wave) * 8 * SG_BK] - OFF / 8), 16, OFF, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gb[q],
                                             (__attribute__((address_space(3))) void*)(&Bs[BUF][(q * 4 + wave) * 8 * SG_BK] - OFF / 8), 16, OFF, 0);
        }
    };
    auto advance = [&](int slabs) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { ga[q] += slabs * SG_BK; gb[q] += slabs * SG_BK; }
    };
    const int swz = (lk >> 1) ^ ((li >> 1) & 7);
    // fragment addresses: row base + 16-byte slot ((2 kq) ^ swz) + the half of it; one register per depth step, the buffer and the
    // row block (m 16 rows = 2 KiB) are immediate offsets of the read
    int fo[4];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) fo[kq] = (((2 * kq) ^ swz) << 1) + (lk & 1) + li * SG_BK;
    const double* const Aw = &As[0][wr * 64 * SG_BK];
    const double* const Bw = &Bs[0][wc * 64 * SG_BK];
    auto frag = [&](auto buf_c, int kq, double (&av)[4], double (&bv)[4]) {          // depth step kq of the slab: columns 4 kq + lk
        constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            av[m] = Aw[BUF * SG_BM * SG_BK + m * 16 * SG_BK + fo[kq]];
            bv[m] = Bw[BUF * SG_BN * SG_BK + m * 16 * SG_BK + fo[kq]];
        }
    };
    sig_f64x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = sig_f64x4{0.0, 0.0, 0.0, 0.0};
    double fa[2][4], fb[2][4];
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    if (nsl > 0) dma(B0{}, B0{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nsl > 0) frag(B0{}, 0, fa[0], fb[0]);
    // one slab out of buffer BUF, the next one (AHEAD slabs behind the pointers) into the other buffer
    auto slab = [&](auto buf_c, auto ahead_c, bool more) {
        constexpr int BUF = decltype(buf_c)::value;
        using OTHER = std::integral_constant<int, BUF ^ 1>;
        if (more) dma(OTHER{}, ahead_c);          // the other buffer: every wave's reads of it were complete at the barrier of the previous slab
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const int cur = kq & 1, nxt = cur ^ 1;
            if (kq < 3) {
                frag(buf_c, kq + 1, fa[nxt], fb[nxt]);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of the next slab have landed
                __syncthreads();                                        // ... and everybody else's; all reads of this slab are done
                if (more) frag(OTHER{}, 0, fa[nxt], fb[nxt]);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][m], fb[cur][n], acc[m][n], 0, 0, 0);
        }
    };
    int s = 0;
    for (; s + 8 <= nsl; s += 8) {               // eight slabs per pointer move: the buffer of each is a compile-time constant
        slab(B0{}, std::integral_constant<int, 1>{}, true);
        slab(B1{}, std::integral_constant<int, 2>{}, true);
        slab(B0{}, std::integral_constant<int, 3>{}, true);
        slab(B1{}, std::integral_constant<int, 4>{}, true);
        slab(B0{}, std::integral_constant<int, 5>{}, true);
        slab(B1{}, std::integral_constant<int, 6>{}, true);
        slab(B0{}, std::integral_constant<int, 7>{}, true);
        slab(B1{}, std::integral_constant<int, 8>{}, s + 8 < nsl);
        advance(8);
    }
    for (; s + 2 <= nsl; s += 2) {               // (s is even here)
        slab(B0{}, std::integral_constant<int, 1>{}, true);
        slab(B1{}, std::integral_constant<int, 2>{}, s + 2 < nsl);
        advance(2);
    }
    if (s < nsl) slab(B0{}, std::integral_constant<int, 1>{}, false);
    double* const P = G.part + int64_t(split) * G.NA * G.NB;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = tile_i + wr * 64 + m * 16 + lk + 4 * r, j = tile_j + wc * 64 + n * 16 + li;
                if (i < G.NA && j < G.NB) P[i * G.NB + j] = acc[m][n][r];
            }
}
#else
__global__ __launch_bounds__(256, 2) void sig_gram_dma_kernel(const SigGramArgs G);
#endif

// mode 0: out[i * so_i + j * so_j] = sum_s part[s][i][j]                                              (general product)
// mode 1: symmetric: tiles bi <= bj hold the sums; out[i][j] = out[j][i]; diag_set: out[i][i] = diag_value (the normalised diagonal is
//         sum_m sigma variances[m] exactly, kernels.py:430-433)
// mode 2: owned entries of rows [r0, r0 + NA) of the symmetric N x N Gram, packed (gpsig_kernel_K_symm_rows_compact): part row i =
//         sequence r0 + i, part column c = sequence (c0 + c) mod N; row j owns column i iff (j - i) mod N < N/2, or == N/2 and (N odd
//         or i < j) -- the rule of seq_emit in seq_args.hpp; out[(j - r0) * (N/2 + 1) + N/2 - (j - i) mod N]
struct SigReduceArgs {
    const double* part; int nsplit; int64_t NA, NB;
    double* out; int64_t so_i, so_j;
    int mode, diag_set; double diag_value;
    int64_t N, r0, c0;
    int compact;          // mode 2: packed rows (1) or full rows of N with only the owned entries written (0)
};
// mode 1 in 32 x 32 blocks of the computed (upper) tiles: partial sums read and the result written along rows, the mirror image written
// along rows too after a transpose through LDS.  grid (blocks per tile side squared, upper tiles), block (32, 8).
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void sig_gram_reduce_sym_kernel(const SigReduceArgs R, int nt) {
    __shared__ double tile[32][33];
    int t = blockIdx.y, bi = 0, rowlen = nt;
    while (t >= rowlen) { t -= rowlen; ++bi; --rowlen; }
    const int bj = bi + t;
    constexpr int SB = SG_BM / 32;
    const int sbi = blockIdx.x / SB, sbj = blockIdx.x - sbi * SB;
    const int64_t i0 = int64_t(bi) * SG_BM + sbi * 32, j0 = int64_t(bj) * SG_BN + sbj * 32;
    const int64_t stride = R.NA * R.NB;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int64_t i = i0 + r, j = j0 + threadIdx.x;
        double s = 0.0;
        if (i < R.NA && j < R.NB) {
            for (int k = 0; k < R.nsplit; ++k) s += R.part[int64_t(k) * stride + i * R.NB + j];
            if (i == j && R.diag_set) s = R.diag_value;
            R.out[i * R.so_i + j * R.so_j] = s;
        }
        tile[r][threadIdx.x] = s;
    }
    if (bi == bj) return;                 // a diagonal tile was computed whole (and a_i . a_j == a_j . a_i bit for bit)
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int64_t j = j0 + r, i = i0 + threadIdx.x;
        if (i < R.NA && j < R.NB) R.out[j * R.so_i + i * R.so_j] = tile[threadIdx.x][r];
    }
}
#else
__global__ void sig_gram_reduce_sym_kernel(const SigReduceArgs R, int nt);
#endif

#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void sig_gram_reduce_kernel(const SigReduceArgs R) {
    const int64_t total = R.NA * R.NB, stride = R.NA * R.NB;
    for (int64_t e = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t i = e / R.NB, j = e - i * R.NB;
        int64_t src = e;
        if (R.mode == 1 && (i / SG_BM) > (j / SG_BN)) src = j * R.NB + i;          // the tile that was computed is the transposed one
        double s = 0.0;
        if (R.mode == 2) {
            const int64_t seq_j = R.r0 + i;
            int64_t seq_i = R.c0 + j;
            if (seq_i >= R.N) seq_i -= R.N;
            const int64_t H = R.N / 2;
            int64_t dlt = seq_j - seq_i;
            if (dlt < 0) dlt += R.N;
            const bool own = dlt < H || (dlt == H && ((R.N & 1) || seq_i < seq_j));
            if (!own) continue;           // (tiles without an owned entry were not computed: SigGramArgs::band)
            for (int k = 0; k < R.nsplit; ++k) s += R.part[int64_t(k) * stride + src];
            if (dlt == 0 && R.diag_set) s = R.diag_value;
            if (R.compact) R.out[i * (H + 1) + H - dlt] = s;
            else R.out[i * R.N + seq_i] = s;
            continue;
        }
        for (int k = 0; k < R.nsplit; ++k) s += R.part[int64_t(k) * stride + src];
        if (R.mode == 1 && i == j && R.diag_set) s = R.diag_value;
        R.out[i * R.so_i + j * R.so_j] = s;
    }
}
#else
__global__ void sig_gram_reduce_kernel(const SigReduceArgs R);
#endif

}  // namespace gpsig
