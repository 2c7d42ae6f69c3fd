// lr_draw_kernels.hpp -- the random objects of a low-rank evaluation drawn ON THE DEVICE: kernels and the state object (the host side
// is gpsig_lr_draw / gpsig_lr_state_* in api.hip).
//
// The reference draws them inside the TensorFlow graph, afresh at every evaluation: the landmarks of Nystrom_map (uniformly,
// without replacement, gpsig/low_rank_calculations.py:12-20, :47-48), the jitter diagonal of :52, and one random projection per
// level >= 2 (lr_hadamard_prod_subsample :104-127 for 'lin'; the very sparse Johnson-Lindenstrauss matrix of :139-149, :177 for
// 'sqrt' / 'log').  Round 2 drew them on the host with NumPy (gpsig_amd/low_rank.py) and uploaded them: 2.4 ms per draw against a
// 2.8 ms evaluation at BASELINE configs[2], most of it rocSOLVER's dsyevd (about 250 small launches for a 50 x 50 matrix) and host
// synchronisations.  Here everything stays on the ctx stream:
//   * a counter-based generator (Philox-4x32-10: the value of stream s at index i is a pure function of (seed, s, i), so the draw
//     does not depend on how the kernels are parallelised),
//   * landmark indices by Floyd's algorithm (a uniformly distributed subset), sorted; the scaled candidates are gathered in place,
//   * the eigendecomposition of the jittered landmark Gram by a cyclic Jacobi iteration in LDS, one workgroup (c <= 64; rocSOLVER,
//     without the trip through the host, beyond that),
//   * the projections column by column: presence of an entry by a Bernoulli(1/s) trial per (row, column), kept in row order by a
//     ballot / prefix-count compaction; N(0, 1) values by Box-Muller.
// What was drawn can be copied out (gpsig_lr_state_export) so that the CPU restatement evaluates the very same random objects.
#pragma once

#include "aux_kernels.hpp"
#include "lr_fused_args.hpp"

namespace gpsig {

// W (c x c, symmetric) += diag(jd)                                                        low_rank_calculations.py:52
__global__ void add_diag_kernel(double* __restrict__ W, const double* __restrict__ jd, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) W[int64_t(i) * c + i] += jd[i];
}
// Sign convention of the eigenvectors (an eigensolver returns each up to sign, the reference's tf.self_adjoint_eig included):
// the component of largest magnitude is made positive (the first such component on ties).  sgn[j] = +-1.
// The level >= 2 features contract coordinate pairs of the whitened features with a fixed random projection, so their
// values -- not their distribution -- depend on these signs; fixing them makes an evaluation a function of its random objects.
__global__ void eig_sign_kernel(const double* __restrict__ Ucm, int c, double* __restrict__ sgn) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c) return;
    double best = 0.0, s = 1.0;
    for (int i = 0; i < c; ++i) {
        const double v = Ucm[int64_t(j) * c + i];
        if (fabs(v) > best) { best = fabs(v); s = v < 0.0 ? -1.0 : 1.0; }
    }
    sgn[j] = s;
}
// Wh[i][j] = sgn[j] U[i][j] / sqrt(ev[j] + jitter), U column-major as dsyevd leaves it    low_rank_calculations.py:56-57, :60
__global__ void whiten_kernel(const double* __restrict__ Ucm, const double* __restrict__ ev, const double* __restrict__ sgn, int c,
                              double jitter, double* __restrict__ Wh) {
    const int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
    if (idx >= int64_t(c) * c) return;
    const int i = int(idx / c), j = int(idx - int64_t(i) * c);
    Wh[idx] = sgn[j] * Ucm[int64_t(j) * c + i] / sqrt(ev[j] + jitter);
}


struct PhiloxKey { uint32_t k0, k1; };

__host__ __device__ inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = uint64_t(0xD2511F53u) * c[0], p1 = uint64_t(0xCD9E8D57u) * c[2];
    const uint32_t n0 = uint32_t(p1 >> 32) ^ c[1] ^ k0, n1 = uint32_t(p1), n2 = uint32_t(p0 >> 32) ^ c[3] ^ k1, n3 = uint32_t(p0);
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
// Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC11): four 32-bit words for counter (index, stream)
__host__ __device__ inline void philox4x32(uint64_t index, uint32_t stream, PhiloxKey key, uint32_t (&out)[4]) {
    uint32_t c[4] = {uint32_t(index), uint32_t(index >> 32), stream, 0u};
    uint32_t k0 = key.k0, k1 = key.k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
// uniform in (0, 1) from 53 random bits
__host__ __device__ inline double philox_u01(uint32_t a, uint32_t b) {
    const uint64_t m = (uint64_t(a >> 5) << 26) | uint64_t(b >> 6);
    return (double(m) + 0.5) * 0x1.0p-53;
}
// uniform integer in [0, n), n < 2^63 (multiply-high of 64 random bits: bias below 2^-64 n)
__host__ __device__ inline uint64_t philox_below(uint32_t a, uint32_t b, uint64_t n) {
    const unsigned __int128 w = (unsigned __int128)((uint64_t(a) << 32) | b) * n;
    return uint64_t(w >> 64);
}

enum : uint32_t { LRS_LANDMARKS = 1, LRS_JITTER = 2, LRS_PRESENT = 16, LRS_VALUE = 32, LRS_LIN = 48 };   // stream ids (+ sketch number)

// c distinct indices out of `total` (Floyd: every c-subset equally likely), ascending.  One wavefront; the list lives in LDS (a thread
// walking a list in global memory pays a memory latency per element: 2.5 ms for 50 landmarks), the membership test is a ballot.
constexpr int LR_DRAW_MAX = 4096;
static __global__ void __launch_bounds__(64) lr_draw_indices_kernel(int64_t total, int c, PhiloxKey key, uint32_t stream, int64_t* __restrict__ idx) {
    __shared__ int64_t lst[LR_DRAW_MAX];
    const int lane = threadIdx.x;
    int n = 0;
    for (int64_t j = total - c; j < total; ++j) {
        uint32_t w[4];
        philox4x32(uint64_t(j), stream, key, w);
        int64_t t = int64_t(philox_below(w[0], w[1], uint64_t(j) + 1));
        bool mine = false;
        for (int k = lane; k < n; k += 64) mine = mine || lst[k] == t;
        if (__ballot(mine) != 0ull) t = j;
        if (lane == 0) {                               // insert, keeping the list ascending
            int k = n;
            while (k > 0 && lst[k - 1] > t) { lst[k] = lst[k - 1]; --k; }
            lst[k] = t;
        }
        ++n;
        __syncthreads();
    }
    for (int k = lane; k < c; k += 64) idx[k] = lst[k];
}

// The landmark candidates are, in this order: the scaled components of Z (ztot rows of d_eff), the scaled observations of X (n1 * l1
// rows), those of X2 (kernels.py:444-446, :562-563).  S[k][fe] = candidate idx[k];  jd[k] = jitter * uniform (low_rank_calculations.py:52).
static __global__ void lr_gather_landmarks_kernel(const int64_t* __restrict__ idx, int c, const double* __restrict__ Z, int64_t ztot,
                                                  const double* __restrict__ X, int64_t n1, int l1, const double* __restrict__ X2, int64_t n2, int l2,
                                                  ScaleParams P, PhiloxKey key, double jitter, double* __restrict__ S, double* __restrict__ jd) {
    const int d_eff = P.d_eff();
    const int64_t total = int64_t(c) * d_eff;
    for (int64_t e = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int k = int(e / d_eff), fe = int(e - int64_t(k) * d_eff);
        int64_t q = idx[k];
        double v;
        if (q < ztot) {                                                              // kernels.py:367-398 on one component
            const int lag = fe / P.d_in, f = fe - lag * P.d_in;
            v = Z[q * d_eff + fe];
            if (P.has_ls) {
                v = v / P.lsv(f);
                if (P.num_lags > 0) v = v * P.gamma[lag];
            }
        } else {
            q -= ztot;
            const bool first = q < n1 * l1;
            if (!first) q -= n1 * l1;
            const int l = first ? l1 : l2;
            const double* A = first ? X : X2;
            const int64_t n = q / l;
            v = scaled_point<double>(A + n * int64_t(l) * P.d_in, l, int(q - n * l), fe, P);
        }
        S[e] = v;
        if (fe == 0) {
            uint32_t w[4];
            philox4x32(uint64_t(k), LRS_JITTER, key, w);
            jd[k] = jitter * philox_u01(w[0], w[1]);
        }
    }
}

// ---- symmetric eigendecomposition, n <= 64, one workgroup: cyclic Jacobi with a round-robin ordering ------------------------------
// A (n x n, row-major, symmetric) in; Ucm (column-major eigenvectors, as dsyevd leaves them) and ev (ascending) out.  Every step rotates
// n/2 disjoint index pairs at once: the angles from the diagonal 2 x 2 blocks, then every 2 x 2 block of A (both rotations at once) and
// the columns of V -- two barriers.
constexpr int LR_JACOBI_MAX = 64;
#ifndef LR_JACOBI_NT
#define LR_JACOBI_NT 1024
#endif
constexpr int LR_JACOBI_THREADS = LR_JACOBI_NT;
constexpr int LR_JACOBI_ITEMS = ((LR_JACOBI_MAX / 2) * (LR_JACOBI_MAX / 2) + LR_JACOBI_MAX * (LR_JACOBI_MAX / 2) + LR_JACOBI_THREADS - 1) / LR_JACOBI_THREADS;
// 1 / sqrt(x), x in [1, 2]: v_rsq_f64 (about 2^-26) and two Newton steps -- the rotation's cosine; any angle close to the annihilating
// one makes the iteration converge, but c^2 + s^2 must be 1 to rounding for V to stay orthogonal
__device__ __forceinline__ double lr_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x, y * y, 1.5);
    y = y * fma(-0.5 * x, y * y, 1.5);
    return y;
}
static __global__ void __launch_bounds__(LR_JACOBI_THREADS) lr_jacobi_eig_kernel(const double* Ain, int n, double* Ucm /* may be Ain */,
                                                                                  double* __restrict__ ev, int* __restrict__ info) {
    extern __shared__ double jsm[];
    const int ld = n + 1;                                // odd-ish stride: a column walk spreads over the banks
    double* A = jsm;                                     // n x ld
    double* V = A + n * ld;                              // n x ld
    double* cs = V + n * ld;                             // 2 per pair
    __shared__ int pq[LR_JACOBI_MAX];                    // (p | q << 8) per pair, -1 for the bye
    __shared__ double offmax, diagmax;
    __shared__ int perm[LR_JACOBI_MAX];
    __shared__ int lone;                                 // odd n: the index that sits this step out
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        A[i * ld + j] = Ain[e];
        V[i * ld + j] = i == j ? 1.0 : 0.0;
    }
    const int np = n + (n & 1), m = np / 2;              // players (one bye when n is odd), pairs per step
    // what this thread updates in a step, decoded once: items [0, m*m) are the 2 x 2 blocks of A (pair k of rows, pair k2 of columns),
    // items [m*m, m*m + n*m) one row of V against one pair; LR_JACOBI_ITEMS per thread at most (m <= 32, n <= 64)
    int it_a[LR_JACOBI_ITEMS], it_b[LR_JACOBI_ITEMS], it_kind[LR_JACOBI_ITEMS];
#pragma unroll
    for (int u = 0; u < LR_JACOBI_ITEMS; ++u) {
        const int e = tid + u * nt;
        it_kind[u] = e < m * m ? 0 : (e < m * m + n * m ? 1 : -1);
        const int f = e < m * m ? e : e - m * m, dv = e < m * m ? m : n;
        it_a[u] = f / dv;
        it_b[u] = f - it_a[u] * dv;
    }
    __syncthreads();
    int sweep = 0;
    for (; sweep < 30; ++sweep) {
        double myoff = 0.0, mydiag = 0.0;                  // per lane over the sweep; reduced once (no same-address LDS atomics per step)
        for (int s = 0; s < np - 1; ++s) {
            if (tid < m) {
                auto player = [&](int pos) { return pos == 0 ? 0 : 1 + (pos - 1 + s) % (np - 1); };
                int p = player(tid), q = player(np - 1 - tid);
                if (p > q) { const int t = p; p = q; q = t; }
                double c = 1.0, sn = 0.0;
                if (q < n) {
                    const double app = A[p * ld + p], aqq = A[q * ld + q], apq = A[p * ld + q];
                    myoff = fmax(myoff, fabs(apq));
                    mydiag = fmax(mydiag, fmax(fabs(app), fabs(aqq)));
                    if (fabs(apq) > 1e-290) {
                        // t = sign(tau) / (|tau| + sqrt(1 + tau^2)), tau = (aqq - app) / (2 apq): hardware reciprocal / root with one
                        // correction each (the angle need not be exact), then c = 1 / sqrt(1 + t^2) to full precision, s = t c
                        const double hd = aqq - app, at = fabs(hd), ab = 2.0 * fabs(apq);
                        // |t| = ab / (at + sqrt(at^2 + ab^2)), scaled by the larger of the two to stay in range
                        const double big = fmax(at, ab), ri = __builtin_amdgcn_rcp(big);
                        const double x = at * ri, y = ab * ri, h2 = fma(x, x, y * y);          // in [1, 2]
                        const double hyp = h2 * lr_rsqrt(h2);
                        double den = x + hyp, rd = __builtin_amdgcn_rcp(den);
                        rd = rd * fma(-den, rd, 2.0);
                        const double tabs = y * rd;
                        const double t = ((hd >= 0.0) == (apq >= 0.0)) ? tabs : -tabs;
                        c = lr_rsqrt(fma(t, t, 1.0));
                        sn = t * c;
                    }
                    pq[tid] = p | (q << 8);
                } else {
                    pq[tid] = -1;
                    lone = p;
                }
                cs[2 * tid] = c; cs[2 * tid + 1] = sn;
            }
            __syncthreads();
            // A <- J^T A J in one pass: the 2 x 2 block (rows of pair k, columns of pair k2) takes pair k's rotation from the left and
            // pair k2's from the right; the blocks of a step are disjoint.  V <- V J alongside.
#pragma unroll
            for (int u = 0; u < LR_JACOBI_ITEMS; ++u) {
                if (it_kind[u] == 0) {
                    const int k = it_a[u], k2 = it_b[u];
                    if (pq[k] < 0 || pq[k2] < 0) continue;      // the index that sits out (odd n): handled below
                    const int p = pq[k] & 255, q = pq[k] >> 8, p2 = pq[k2] & 255, q2 = pq[k2] >> 8;
                    const double c = cs[2 * k], sn = cs[2 * k + 1], c2 = cs[2 * k2], s2 = cs[2 * k2 + 1];
                    const double a00 = A[p * ld + p2], a01 = A[p * ld + q2], a10 = A[q * ld + p2], a11 = A[q * ld + q2];
                    const double b00 = c * a00 - sn * a10, b01 = c * a01 - sn * a11;        // rows: [p; q] <- [[c -s] [s c]] [p; q]
                    const double b10 = sn * a00 + c * a10, b11 = sn * a01 + c * a11;
                    A[p * ld + p2] = c2 * b00 - s2 * b01;                                   // columns: [p2 q2] <- [p2 q2] [[c2 s2] [-s2 c2]]
                    A[p * ld + q2] = s2 * b00 + c2 * b01;
                    A[q * ld + p2] = c2 * b10 - s2 * b11;
                    A[q * ld + q2] = s2 * b10 + c2 * b11;
                } else if (it_kind[u] == 1) {
                    const int k = it_a[u], i = it_b[u];
                    if (pq[k] < 0) continue;
                    const int p = pq[k] & 255, q = pq[k] >> 8;
                    const double c = cs[2 * k], sn = cs[2 * k + 1];
                    const double vp = V[i * ld + p], vq = V[i * ld + q];
                    V[i * ld + p] = c * vp - sn * vq;
                    V[i * ld + q] = sn * vp + c * vq;
                }
            }
            if ((n & 1) && tid < m && pq[tid] >= 0) {
                // odd n: the index that sits this step out still meets every pair -- as a single row (rotated from the right) and,
                // by symmetry, a single column; nobody else touches that row and column in this step
                const int p = pq[tid] & 255, q = pq[tid] >> 8, lo = lone;
                const double c = cs[2 * tid], sn = cs[2 * tid + 1];
                const double rp = A[lo * ld + p], rq = A[lo * ld + q];
                const double np_ = c * rp - sn * rq, nq_ = sn * rp + c * rq;
                A[lo * ld + p] = np_; A[lo * ld + q] = nq_;
                A[p * ld + lo] = np_; A[q * ld + lo] = nq_;
            }
            __syncthreads();
        }
        // converged when a whole sweep met no off-diagonal entry above the rounding level of the diagonal (a rotation leaves its
        // own entry at about 1e-17 of the diagonal rather than at zero, so that is where the entries settle)
        if (tid < 64) {                                     // the pairs live in the first wavefront (m <= 32)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                myoff = fmax(myoff, __shfl_xor(myoff, o, 64));
                mydiag = fmax(mydiag, __shfl_xor(mydiag, o, 64));
            }
            if (tid == 0) { offmax = myoff; diagmax = mydiag; }
        }
        __syncthreads();
        const bool done = offmax <= 1e-15 * diagmax;
        __syncthreads();
        if (done) break;
    }
    if (tid == 0) {                                        // ascending order, as LAPACK / tf.self_adjoint_eig return it
        for (int j = 0; j < n; ++j) perm[j] = j;
        for (int j = 1; j < n; ++j) {
            const int pj = perm[j];
            const double vj = A[pj * ld + pj];
            int k = j;
            while (k > 0 && A[perm[k - 1] * ld + perm[k - 1]] > vj) { perm[k] = perm[k - 1]; --k; }
            perm[k] = pj;
        }
        *info = sweep >= 30 ? 1 : 0;
        info[2] = sweep;                               // sweeps taken (diagnostics)
    }
    __syncthreads();
    for (int j = tid; j < n; j += nt) ev[j] = A[perm[j] * ld + perm[j]];
    for (int e = tid; e < n * n; e += nt) {
        const int j = e / n, i = e - j * n;
        Ucm[e] = V[i * ld + perm[j]];
    }
}

// ---- sparse projections -----------------------------------------------------------------------------------------------------------
// 'sqrt' / 'log': entry (row, col) of the D x r matrix is present with probability 1/s.  One wavefront per column; rows in order.
// fill == 0: counts[col] only.  fill == 1: entries written at colptr[col] .. in row order, values N(0, 1) sqrt(s / r).
static __global__ void __launch_bounds__(64) lr_draw_sparse_kernel(int64_t D, int r, int k1, double inv_s, double scale, PhiloxKey key, uint32_t sk,
                                                                   int fill, int32_t cap, int32_t* __restrict__ counts,
                                                                   const int32_t* __restrict__ colptr, int32_t* __restrict__ i1, int32_t* __restrict__ i2,
                                                                   double* __restrict__ val, LrEntry* __restrict__ ent) {
    const int col = blockIdx.x, lane = threadIdx.x;
    int32_t at = fill ? colptr[col] : 0;
    for (int64_t row0 = 0; row0 < D; row0 += 64) {
        const int64_t row = row0 + lane;
        uint32_t w[4];
        philox4x32(uint64_t(col) * uint64_t(D) + uint64_t(row), LRS_PRESENT + sk, key, w);
        const bool present = row < D && philox_u01(w[0], w[1]) < inv_s;
        const unsigned long long mask = __ballot(present);
        if (fill && present) {
            const int32_t o = at + __popcll(mask & ((1ull << lane) - 1ull));
            if (o < cap) {
                // Box-Muller on the other two words of the same counter
                const double u1 = philox_u01(w[2], w[3]);
                uint32_t w2[4];
                philox4x32(uint64_t(col) * uint64_t(D) + uint64_t(row), LRS_VALUE + sk, key, w2);
                const double u2 = philox_u01(w2[0], w2[1]);
                const double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2) * scale;
                const int32_t a = int32_t(row % k1), b = int32_t(row / k1);
                i1[o] = a; i2[o] = b; val[o] = g;
                ent[o] = LrEntry{g, a, b};
            }
        }
        at += __popcll(mask);
    }
    if (!fill && lane == 0) counts[col] = at;
}
// colptr = exclusive prefix of counts, clamped to the capacity (nnz = colptr[r]).  One wavefront, 64 columns per pass.
static __global__ void __launch_bounds__(64) lr_colptr_kernel(const int32_t* __restrict__ counts, int r, int32_t cap, int32_t* __restrict__ colptr,
                                                              int* __restrict__ overflow) {
    const int lane = threadIdx.x;
    int64_t base = 0;
    for (int j0 = 0; j0 < r; j0 += 64) {
        const int j = j0 + lane;
        const int64_t v = j < r ? counts[j] : 0;
        int64_t inc = v;                                   // inclusive prefix over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        const int64_t excl = base + inc - v;
        if (j < r) colptr[j] = int32_t(excl < cap ? excl : cap);
        base += __shfl(inc, 63, 64);
    }
    if (lane == 0) {
        colptr[r] = int32_t(base < cap ? base : cap);
        if (base > cap) *overflow = 1;
    }
}
// 'lin' (low_rank_calculations.py:104-127): r distinct coordinate pairs with Rademacher signs, one per output column.  One wavefront.
static __global__ void __launch_bounds__(64) lr_draw_lin_kernel(int64_t D, int r, int k1, PhiloxKey key, uint32_t sk, int32_t* __restrict__ colptr,
                                                                int32_t* __restrict__ i1, int32_t* __restrict__ i2, double* __restrict__ val,
                                                                LrEntry* __restrict__ ent) {
    __shared__ int64_t lst[LR_DRAW_MAX];
    __shared__ signed char sg[LR_DRAW_MAX];
    const int lane = threadIdx.x;
    int n = 0;
    for (int64_t j = D - r; j < D; ++j) {                  // Floyd; kept in draw order (any order is as good)
        uint32_t w[4];
        philox4x32(uint64_t(j), LRS_LIN + sk, key, w);
        int64_t t = int64_t(philox_below(w[0], w[1], uint64_t(j) + 1));
        bool mine = false;
        for (int k = lane; k < n; k += 64) mine = mine || lst[k] == t;
        if (__ballot(mine) != 0ull) t = j;
        if (lane == 0) { lst[n] = t; sg[n] = (w[2] & 1u) ? 1 : -1; }
        ++n;
        __syncthreads();
    }
    for (int k = lane; k < r; k += 64) {
        const double sgn = double(sg[k]);
        const int32_t a = int32_t(lst[k] % k1), b = int32_t(lst[k] / k1);
        colptr[k] = k;
        i1[k] = a; i2[k] = b; val[k] = sgn;
        ent[k] = LrEntry{sgn, a, b};
    }
    if (lane == 0) colptr[r] = r;
}

static __global__ void lr_transpose_kernel(const double* __restrict__ A, int n, double* __restrict__ At) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n * n) At[(e % n) * n + e / n] = A[e];
}

// a failed draw poisons the whitening (and with it every feature computed from the state): info[0] eigensolver, info[1] capacity
__global__ void lr_poison_kernel(const int* __restrict__ info, int c, double* __restrict__ Wh, double* __restrict__ WhT) {
    if (info[0] == 0 && info[1] == 0) return;
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < int64_t(c) * c) {
        const double nan = __longlong_as_double(0x7ff8000000000000ll);
        Wh[i] = nan; WhT[i] = nan;
    }
}

}  // namespace gpsig

struct gpsig_lr_state {
    gpsig_ctx* ctx = nullptr;       // NULL once the context was destroyed (gpsig_ctx_destroy detaches its states)
    int device = 0;                 // ... and then the device the block lives on
    int c = 0, d_eff = 0, r = 0, nsk = 0, sparsity = 0;
    void* block = nullptr;
    size_t bytes = 0;
    int64_t* idx = nullptr;
    double *S = nullptr, *jd = nullptr, *W = nullptr, *Wh = nullptr, *WhT = nullptr, *ev = nullptr, *work = nullptr;
    int* info = nullptr;            // [0] eigensolver, [1] projection capacity exceeded
    struct Sk {
        int k1 = 0, k2 = 0;
        int32_t cap = 0;
        int32_t *counts = nullptr, *colptr = nullptr, *i1 = nullptr, *i2 = nullptr;
        double* val = nullptr;
        gpsig::LrEntry* ent = nullptr;
    } sk[gpsig::LR_FUSED_MAX_SKETCHES];
};

