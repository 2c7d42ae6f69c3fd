// seq_core.hpp -- per-lane arithmetic of the sequence-vs-sequence signature-kernel recursion.
//
// Shared, verbatim, by the gfx950 kernel (seq_gram_kernel.hpp) and by the host-side lock-step wave
// emulator that the CPU test-suite runs (tests/emu/).  Nothing here touches memory spaces or lanes:
// cross-lane inputs arrive in NbrIn, the x-side row arrives as a register array.
//
// What it computes (first-order algorithm, gpsig/signature_algs.py:8-35).  For one pair (x, y) with
// increment lattice dM[a][b] (signature_algs.py:26) the reference evaluates, level by level over the
// whole lattice,  R_1 = dM,  R_m = dM * excumsum_b(excumsum_a(R_{m-1})),  K_m = sum R_m.
// Here the same numbers come from ONE sweep over lattice rows a that carries, per level m, the
// inclusive 2-D prefix  Q_m[a][b] = sum_{a'<=a, b'<=b} R_m[a'][b']  of the previous row only:
//     s_m[a][b] = s_m[a][b-1] + dM[a][b] * Q_{m-1}[a-1][b-1]      (row prefix of R_m;  Q_0 == 1)
//     Q_m[a][b] = Q_m[a-1][b] + s_m[a][b]
//     K_m       = Q_m[last][last]
// i.e. two fp64 instructions per lattice cell and level (one FMA, one add) instead of the
// reference's five full-lattice passes, with O(M * L2) state instead of O(L1 * L2).
//
// Lane mapping.  A pair occupies G consecutive lanes; lane `lam` owns C consecutive lattice columns
// (record rows C*lam .. C*lam+C-1 of the y side).  Row prefixes cross lanes through a carry that is
// handed to the next lane ONE STEP LATER (lane lam works on lattice row t-lam at step t), so no
// log-step scan is needed: each step a lane reads its left neighbour's end-of-chunk row prefix
// (`cin`); the neighbour's last-column Q it also needs is tracked locally as a ghost column (SeqLane::qg).
#pragma once

#include <cmath>

#include "fast_exp.hpp"

#ifndef SEQ_EXP256
#define SEQ_EXP256 1      // float64 RBF pair kernels: 1 = the 256-entry exp table with the degree-4 tail, 0 = 64 entries / degree 5
                          // (same box, alternating, BASELINE configs[1] RBF: 45.44 -> 44.04 ms; profiles/r02_ab_exp256.txt)
#endif

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GPSIG_HD __host__ __device__ __forceinline__
#else
#define GPSIG_HD inline
#endif

namespace gpsig {

constexpr int SEQ_ETAB_N = SEQ_EXP256 ? EXP_TAB256_N : EXP_TAB_N;
constexpr double SEQ_RBF_PRESCALE = SEQ_EXP256 ? EXP_PRESCALE256 : EXP_PRESCALE;

// values equal enum gpsig_base_kernel in include/gpsig_hip.h
enum : int { BASE_LINEAR = 0, BASE_RBF = 1, BASE_COSINE = 2, BASE_POLY = 3, BASE_MIX = 4,
             BASE_MATERN12 = 5, BASE_MATERN32 = 6, BASE_MATERN52 = 7, BASE_SPECTRAL = 8 };

// how dM is produced
enum : int {
    MODE_INC = 0,      // dM[a][b] = <xrow_a, yrow_b>      (linear base kernel: rows are increments, or points if difference=False)
    MODE_PT_DIFF = 1,  // dM = double increment of kappa(x_a, y_b) over point rows (signature_algs.py:26)
    MODE_PT_NODIFF = 2 // dM[a][b] = kappa(x_a, y_b)        (difference=False, non-linear base kernel)
};

// exp of a non-positive argument.  float32 on the device: v_exp_f32(x * log2 e), about 1e-6 relative over the range a
// kernel value can matter in (the float32 tolerance is 1e-4); float64 and the host: the library exp.
GPSIG_HD double kexp(double x) { return exp(x); }
GPSIG_HD float kexp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __expf(x);
#else
    return expf(x);
#endif
}

// SignatureSpectral's state-space kernel (gpsig/kernels.py:921-942):
//     kappa(x, y) = sum_q alpha_q * E_q * cos(2 pi <omega_q, x - y>),   E_q = exp(-|gamma_q (x - y)|^2 / 2)   ('rbf' family)
//                                                                        or  exp(-|gamma_q (x - y)| / 2)     ('exp' family)
// 'mixed' (:932-936 reference undefined names and a sign error there; the evident intent is taken): the first floor(Q/2)
// components Gaussian, the rest exponential.  It is not a function of (<x,y>, |x|^2, |y|^2), so it takes the points; the
// wavefront kernel (seq_step below) does not carry it -- sequence-vs-sequence evaluations with this kernel go through the
// one-pair-per-thread kernel of aux_kernels.hpp.
// Table layout (doubles): alpha[Q], omega[Q][SPECTRAL_STRIDE], gamma[Q][SPECTRAL_STRIDE], zero beyond the d features.
enum : int { SPECTRAL_RBF = 0, SPECTRAL_EXP = 1, SPECTRAL_MIXED = 2, SPECTRAL_STRIDE = 32 };
template <typename T, class FX, class FY>
GPSIG_HD T spectral_eval(const double* __restrict__ tab, int Q, int family, int d, FX&& xf, FY&& yf) {
    constexpr int DT = SPECTRAL_STRIDE;
    T acc = T(0);
    for (int q = 0; q < Q; ++q) {
        const double* om = tab + Q + q * DT;
        const double* ga = tab + Q + Q * DT + q * DT;
        T w1 = T(0), w2 = T(0);
        for (int f = 0; f < d; ++f) {
            const T diff = xf(f) - yf(f);
            const T gd = T(ga[f]) * diff;
            w1 = fma(gd, gd, w1);
            w2 = fma(T(om[f]), diff, w2);
        }
        const bool gauss = family == SPECTRAL_RBF || (family == SPECTRAL_MIXED && q < Q / 2);
        const T env = gauss ? kexp(-w1 / 2) : kexp(-sqrt(w1) / 2);
        acc = fma(T(tab[q]) * env, cos(T(6.283185307179586476925) * w2), acc);
    }
    return acc;
}

// b^p for SignaturePoly (kernels.py:844-848: tf.pow(x y^T + gamma, degree)).  The degree is an integer in every use the reference makes of
// it (default 3): a small whole exponent is taken by repeated squaring -- three multiplications at most for p <= 8 against the ~150
// instructions of the library's float64 pow, which made the polynomial kernel's evaluations 3-6 times the RBF kernel's (round 4:
// tools/bench_train_paths.py); any other exponent goes to pow.  The branch is uniform: p is a kernel argument.
// (The library pow stays OUT OF LINE: inlined into the kernels that switch over the base kernel at run time it had been their register hog --
// seq_gram_kernel<double, 16, 2, 4, 8> 159 registers with it, 193 once this helper joined it: three wavefronts per SIMD -> two -- and a
// one-register cliff in the low-rank feature kernel)
template <typename T>
GPSIG_HD __attribute__((noinline)) T poly_pow_general(T b, T p) { return pow(b, p); }

template <typename T>
GPSIG_HD T poly_pow(T b, T p) {
    const int n = int(p);
    if (T(n) == p && n >= 0 && n <= 8) {
        T r = (n & 1) ? b : T(1), x = b * b;
        if (n & 2) r *= x;
        x *= x;
        if (n & 4) r *= x;
        if (n & 8) r *= x * x;
        return r;
    }
    return poly_pow_general(b, p);
}

// Static kernel on R^d from the inner product and the two squared norms (gpsig/kernels.py:765-781, 799-993).
template <typename T>
GPSIG_HD T base_eval(int kind, T inner, T xs, T ys, T p0, T p1) {
    switch (kind) {
        case BASE_LINEAR: return inner;                                            // :799-806
        case BASE_COSINE: return inner / (sqrt(xs) * sqrt(ys));                     // :820-828
        case BASE_POLY: return poly_pow(inner + p0, p1);                            // :844-848
        default: break;
    }
    const T dist = fma(T(-2), inner, xs + ys);                                      // _square_dist :765-776
    if (kind == BASE_RBF) return kexp(-dist / 2);                                    // :862-864
    if (kind == BASE_MIX) return p0 * kexp(-dist / 2) + (T(1) - p0) * inner;         // :881-892
    const T r = sqrt(fmax(dist, T(1e-40)));                                         // _euclid_dist :779-781
    if (kind == BASE_MATERN12) return kexp(-r);                                      // :955-958
    if (kind == BASE_MATERN32) {                                                    // :974-977
        const T c = T(1.7320508075688772935);
        return (T(1) + c * r) * kexp(-c * r);
    }
    const T c = T(2.2360679774997896964);                                           // :991-993
    return (T(1) + c * r + T(5.0 / 3.0) * (r * r)) * kexp(-c * r);
}

// The same for NV inner products at once, in place: v[i] (inner product) -> kappa.  a2[i] is the squared norm
// on the per-value side, b2 the squared norm of the shared point.  The switch on the kernel family sits OUTSIDE
// the unrolled loops: one wave-uniform branch per call instead of one per value.
template <typename T, int NV>
GPSIG_HD void base_eval_n(int kind, T (&v)[NV], const T (&a2)[NV], T b2, T p0, T p1, int nvalid = NV) {
    switch (kind) {
        case BASE_LINEAR: return;
        case BASE_COSINE: {
            const T sb = sqrt(b2);
#pragma unroll
            for (int i = 0; i < NV; ++i) if (i < nvalid) v[i] = v[i] / (sb * sqrt(a2[i]));
            return;
        }
        case BASE_POLY:
#pragma unroll
            for (int i = 0; i < NV; ++i) if (i < nvalid) v[i] = poly_pow(v[i] + p0, p1);
            return;
        case BASE_RBF:
#pragma unroll
            for (int i = 0; i < NV; ++i) if (i < nvalid) v[i] = kexp(-fma(T(-2), v[i], b2 + a2[i]) / 2);
            return;
        case BASE_MIX:
#pragma unroll
            for (int i = 0; i < NV; ++i) if (i < nvalid) v[i] = p0 * kexp(-fma(T(-2), v[i], b2 + a2[i]) / 2) + (T(1) - p0) * v[i];
            return;
        case BASE_MATERN12:
#pragma unroll
            for (int i = 0; i < NV; ++i) if (i < nvalid) v[i] = kexp(-sqrt(fmax(fma(T(-2), v[i], b2 + a2[i]), T(1e-40))));
            return;
        case BASE_MATERN32: {
            const T c = T(1.7320508075688772935);
#pragma unroll
            for (int i = 0; i < NV; ++i) if (i < nvalid) {
                const T r = sqrt(fmax(fma(T(-2), v[i], b2 + a2[i]), T(1e-40)));
                v[i] = (T(1) + c * r) * kexp(-c * r);
            }
            return;
        }
        default: {
            const T c = T(2.2360679774997896964);
#pragma unroll
            for (int i = 0; i < NV; ++i) if (i < nvalid) {
                const T r = sqrt(fmax(fma(T(-2), v[i], b2 + a2[i]), T(1e-40)));
                v[i] = (T(1) + c * r + T(5.0 / 3.0) * (r * r)) * kexp(-c * r);
            }
            return;
        }
    }
}

template <typename T, int C, int D, int MMAX, int MODE>
struct SeqLane {
    static constexpr int NQ = MMAX > 1 ? MMAX - 1 : 1;
    T y[C][D];      // y-side record rows owned by this lane (increments, or points)
    T q[NQ][C];     // Q_m for levels m = 1 .. M-1 (index m-1), current as of the last processed row
    T qg[NQ];       // ghost column: Q_m of the LEFT neighbour's last column, as of the last processed row.  It is advanced
                    // with the carry received from that neighbour (qg += cin), which is bit for bit the neighbour's own update,
                    // so the diagonal term Q_{m-1}[a-1][b-1] of a lane's first column costs one add instead of a hand-over
    T s[MMAX];      // end-of-chunk row prefix of R_m for the last processed row (the carry handed right)
    T ktop;         // running K_M (only the row totals of the top level are needed)
    T keep;         // 1, or 0 at a step that opens a new pair: the accumulators are updated as  acc = acc * keep + increment,
                    // which is acc + increment bit for bit while keep == 1 and clears them, at no cost, in the one step where the
                    // increment is zero anyway (row 0 of an x).  It replaces reset() at pair boundaries on the GPU: there the
                    // boundary block runs for 4 lanes of 64 in 16 of every R1 steps, and clearing 4M+1 registers in it was ~8 %
                    // of the headline kernel's instructions.  (0 * inf is NaN: the kernel falls back to reset() when ktop is not finite.)
    // point modes only
    T y2[C];        // |y|^2 per owned column
    T eprev[C];     // kappa(previous x point, owned column r) - kappa(previous x point, column r - 1): the column difference of the
                    // previous row, kept so that the double increment of signature_algs.py:26 costs two subtractions per cell
                    // instead of three (the same operations on the same operands as (k[a][b] - k[a][b-1]) - (k[a-1][b] - k[a-1][b-1]):
                    // bit-identical)
    T klast;        // kappa(previous x point, last owned column): what the right neighbour reads as its column -1
    static constexpr bool HIGHER_ORDER = false;

    // Pair boundary.  Only the accumulators are cleared: s[] holds the hand-over words that the RIGHT neighbour
    // still has to read during this very step (it is one lattice row behind); they are rewritten by this lane's
    // own step before anyone reads them again.
    GPSIG_HD void reset() {
#pragma unroll
        for (int m = 0; m < NQ; ++m) {
#pragma unroll
            for (int r = 0; r < C; ++r) q[m][r] = T(0);
            qg[m] = T(0);
        }
        ktop = T(0);
    }
    GPSIG_HD void init() {
        reset();
        keep = T(1);
#pragma unroll
        for (int m = 0; m < MMAX; ++m) s[m] = T(0);
#pragma unroll
        for (int r = 0; r < C; ++r) { eprev[r] = T(0); y2[r] = T(0); }
        klast = T(0);
    }
    // K_m for m = 1..M as seen by the LAST lane of the pair's group
    GPSIG_HD T level_value(int m, int M) const {
        T v = ktop;
#pragma unroll
        for (int k = 0; k < NQ; ++k)
            if (k == m - 1 && m < M) v = q[k][C - 1];
        return v;
    }
};

// Cross-lane inputs are fetched through a policy object `Nbr` with three members, each returning the
// LEFT neighbour's copy of one of this lane's own state words as of the end of the previous step:
//     T cin(int m)      left neighbour's s[m]
//     T kleft()         left neighbour's klast           (MODE_PT_DIFF)
//     T win(int m, int r)  left neighbour's w[m][r]      (higher-order lanes)
// (zero for the first lane of a pair group).  On the GPU they are DPP row/wave shifts issued right
// where the value is consumed -- legal because s[m] / kprev are only overwritten later in
// the same step -- which keeps the 2M+1 shifted words out of the live register set.  The CPU
// emulator serves them from a snapshot (NbrSnapshot) taken before any lane of the wave has stepped.
template <typename T, int MMAX>
struct NbrSnapshot {
    static constexpr int NQ = MMAX > 1 ? MMAX - 1 : 1;
    T s[MMAX];
    T klast;
    T w[NQ][8];    // higher-order lanes only
    GPSIG_HD T cin(int m) const { return s[m]; }
    GPSIG_HD T kleft() const { return klast; }
    GPSIG_HD T win(int m, int r) const { return w[m][r]; }
};

namespace detail {
template <int MI, typename T, int C, int D, int MMAX, int MODE, class Nbr>
GPSIG_HD void seq_level(SeqLane<T, C, D, MMAX, MODE>& L, const Nbr& nbr, const T (&dm)[C], int M) {
    if (MI < M) {                       // wave-uniform (compile-time when M is)
        const T cin = nbr.cin(MI);      // left neighbour's end-of-chunk row prefix of R_m for this lattice row
        T sm = cin;
        if (MI == M - 1) {              // top level: accumulate the row total only
            if constexpr (MI == 0) {
#pragma unroll
                for (int r = 0; r < C; ++r) sm += dm[r];
            } else {
                sm = fma(dm[0], L.qg[MI - 1], sm);
#pragma unroll
                for (int r = 1; r < C; ++r) sm = fma(dm[r], L.q[MI - 1][r - 1], sm);
            }
            L.ktop = fma(L.ktop, L.keep, sm);
        } else if constexpr (MI < MMAX - 1) {
            if constexpr (MI == 0) {
#pragma unroll
                for (int r = 0; r < C; ++r) { sm += dm[r]; L.q[0][r] = fma(L.q[0][r], L.keep, sm); }
            } else {
                sm = fma(dm[0], L.qg[MI - 1], sm);
                L.q[MI][0] = fma(L.q[MI][0], L.keep, sm);
#pragma unroll
                for (int r = 1; r < C; ++r) { sm = fma(dm[r], L.q[MI - 1][r - 1], sm); L.q[MI][r] = fma(L.q[MI][r], L.keep, sm); }
            }
            L.qg[MI] = fma(L.qg[MI], L.keep, cin);   // the ghost column moves past this row (read by level MI+2 ... already done: descending order)
        }
        L.s[MI] = sm;
    }
    if constexpr (MI > 0) seq_level<MI - 1>(L, nbr, dm, M);   // descending: level m+1 reads Q_m before level m updates it
}
}  // namespace detail

// One lattice row for one lane, given dM for the lane's C columns.
template <typename T, int C, int D, int MMAX, int MODE, class Nbr>
GPSIG_HD void seq_recursion(SeqLane<T, C, D, MMAX, MODE>& L, const Nbr& nbr, const T (&dm)[C], int M) {
    detail::seq_level<MMAX - 1>(L, nbr, dm, M);
}

// Full first-order step.  xr: the x-side record row for this step.  dummy: this step is not a lattice row of the
// current pair (pair boundary / lane outside its active window): contributes nothing.
// [rlo, rhi): owned columns that are real lattice columns (point modes; MODE_INC relies on zero rows).
template <typename T, int C, int D, int MMAX, int MODE, class Nbr>
GPSIG_HD void seq_step(SeqLane<T, C, D, MMAX, MODE>& L, const Nbr& nbr, const T (&xr)[D], int M, int /*order*/,
                       bool dummy, int rlo, int rhi, int kind, T p0, T p1);

// =====================================================================================================
// Higher-order algorithm (gpsig/signature_algs.py:37-74), same lane mapping and skew.
//
// The reference keeps, per level, a d x d grid (d = min(level, order)) of full lattices R[r][s] indexed by how
// often the last x index (r+1 times) and the last y index (s+1 times) have been repeated:
//     R_m[0][0] = dM * excumsum_ab( sum_{r,s} R_{m-1}[r][s] )                       (:64)
//     R_m[0][s] = dM/(s+1) * excumsum_a( sum_r R_{m-1}[r][s-1] )                    (:66)
//     R_m[r][0] = dM/(r+1) * excumsum_b( sum_s R_{m-1}[r-1][s] )                    (:67)
//     R_m[r][s] = dM/((r+1)(s+1)) * R_{m-1}[r-1][s-1]                               (:69)
//     K_m = sum_{a,b} sum_{r,s} R_m[r][s]                                           (:71)
// In the row sweep that becomes, per level m < M and per owned column:  QT_m (inclusive 2-D prefix of the grid
// total, as in the first-order case), PC_{m,s} (column prefix, i.e. running sum over rows, of sum_r R_m[r][s]),
// and per row the running row prefixes of sum_s R_m[r][s], whose chunk-end values `w` join `s` as
// hand-over words to the right neighbour.  Levels are processed in ascending order inside a step because level m
// needs level m-1's values of the SAME lattice row; level m-1's prefixes are advanced past the row only after
// level m has read their previous-row values.
template <typename T, int C, int D, int MMAX, int OMAX, int MODE>
struct SeqLaneHO {
    static constexpr int NQ = MMAX > 1 ? MMAX - 1 : 1;
    static constexpr int NO = OMAX > 1 ? OMAX - 1 : 1;
    static constexpr bool HIGHER_ORDER = true;
    T y[C][D];
    T q[NQ][C];        // QT_m
    T qg[NQ];          // ghost column of QT_m (see SeqLane::qg)
    T s[MMAX];         // chunk-end row prefix of the level total                   (hand-over)
    T pc[NQ][NO][C];   // PC_{m,s}
    T w[NQ][NO];       // chunk-end row prefix of sum_s R_m[r][s] for r < order-1   (hand-over)
    T ktop;
    T y2[C], eprev[C], klast;      // as in SeqLane

    GPSIG_HD void reset() {
#pragma unroll
        for (int m = 0; m < NQ; ++m)
#pragma unroll
            for (int r = 0; r < C; ++r) {
                q[m][r] = T(0);
#pragma unroll
                for (int o = 0; o < NO; ++o) pc[m][o][r] = T(0);
            }
#pragma unroll
        for (int m = 0; m < NQ; ++m) qg[m] = T(0);
        ktop = T(0);
    }
    GPSIG_HD void init() {
        reset();
#pragma unroll
        for (int m = 0; m < NQ; ++m) {
#pragma unroll
            for (int o = 0; o < NO; ++o) w[m][o] = T(0);
        }
#pragma unroll
        for (int m = 0; m < MMAX; ++m) s[m] = T(0);
#pragma unroll
        for (int r = 0; r < C; ++r) { eprev[r] = T(0); y2[r] = T(0); }
        klast = T(0);
    }
    GPSIG_HD T level_value(int m, int M) const {
        T v = ktop;
#pragma unroll
        for (int k = 0; k < NQ; ++k)
            if (k == m - 1 && m < M) v = q[k][C - 1];
        return v;
    }
};

namespace detail {
// level LV (1-based, compile time) of the higher-order step; Rp = level LV-1's grid values of this lattice row
template <int LV, typename T, int C, int D, int MMAX, int OMAX, int MODE, class Nbr>
GPSIG_HD void seq_ho_level(SeqLaneHO<T, C, D, MMAX, OMAX, MODE>& L, const Nbr& nbr, const T (&dm)[C], int M, int order,
                           T (&Rp)[OMAX][OMAX][C]) {
    if (LV <= M) {
        constexpr int DC = LV < OMAX ? LV : OMAX;                // static bounds of this / the previous level's grid
        constexpr int DP = (LV - 1) < OMAX ? (LV - 1) : OMAX;
        const int dcur = LV < order ? LV : order;                // signature_algs.py:62
        const int dprev = (LV - 1) < order ? (LV - 1) : order;
        T Rc[OMAX][OMAX][C];
        if constexpr (LV == 1) {
#pragma unroll
            for (int c = 0; c < C; ++c) Rc[0][0][c] = dm[c];                                        // :58-60
        } else {
            constexpr int MI = LV - 2;                           // state index of level LV-1
            T tot[C], cs[DP > 0 ? DP : 1][C], rs[DP > 0 ? DP : 1][C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                tot[c] = T(0);
#pragma unroll
                for (int a = 0; a < DP; ++a) { cs[a][c] = T(0); rs[a][c] = T(0); }
#pragma unroll
                for (int r = 0; r < DP; ++r)
#pragma unroll
                    for (int sx = 0; sx < DP; ++sx)
                        if (r < dprev && sx < dprev) { tot[c] += Rp[r][sx][c]; cs[sx][c] += Rp[r][sx][c]; rs[r][c] += Rp[r][sx][c]; }
            }
            T wrun[DC > 1 ? DC - 1 : 1];
#pragma unroll
            for (int r = 0; r < DC - 1; ++r) wrun[r] = (r < dcur - 1) ? nbr.win(MI, r) : T(0);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const T qd = (c == 0) ? L.qg[MI] : L.q[MI][c == 0 ? 0 : c - 1];
                Rc[0][0][c] = dm[c] * qd;                                                             // :64
#pragma unroll
                for (int sx = 1; sx < DC; ++sx)
                    if (sx < dcur) Rc[0][sx][c] = (dm[c] * (T(1) / T(sx + 1))) * L.pc[MI][sx - 1][c];  // :66
#pragma unroll
                for (int r = 1; r < DC; ++r)
                    if (r < dcur) Rc[r][0][c] = (dm[c] * (T(1) / T(r + 1))) * wrun[r - 1];             // :67
#pragma unroll
                for (int r = 1; r < DC; ++r)
#pragma unroll
                    for (int sx = 1; sx < DC; ++sx)
                        if (r < dcur && sx < dcur)
                            Rc[r][sx][c] = (dm[c] * (T(1) / (T(r + 1) * T(sx + 1)))) * Rp[r - 1][sx - 1][c];   // :69
#pragma unroll
                for (int r = 0; r < DC - 1; ++r)
                    if (r < dcur - 1) wrun[r] += rs[r][c];
            }
            // level LV-1's prefixes move past this lattice row now that their previous-row values have been used
            const T cin = nbr.cin(MI);
            T srun = cin;
            L.qg[MI] += cin;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                srun += tot[c];
                L.q[MI][c] += srun;
#pragma unroll
                for (int sx = 0; sx < DC - 1; ++sx)
                    if (sx < dcur - 1) L.pc[MI][sx][c] += cs[sx][c];
            }
            L.s[MI] = srun;
#pragma unroll
            for (int r = 0; r < DC - 1; ++r)
                if (r < dcur - 1) L.w[MI][r] = wrun[r];
        }
        if (LV == M) {              // top level: only the lattice total is needed (signature_algs.py:71)
            T srun = nbr.cin(LV - 1);
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int r = 0; r < DC; ++r)
#pragma unroll
                    for (int sx = 0; sx < DC; ++sx)
                        if (r < dcur && sx < dcur) srun += Rc[r][sx][c];
            L.s[LV - 1] = srun;
            L.ktop += srun;
        }
        if constexpr (LV < MMAX) {
            T Rn[OMAX][OMAX][C];
#pragma unroll
            for (int r = 0; r < DC; ++r)
#pragma unroll
                for (int sx = 0; sx < DC; ++sx)
#pragma unroll
                    for (int c = 0; c < C; ++c) Rn[r][sx][c] = Rc[r][sx][c];
            seq_ho_level<LV + 1>(L, nbr, dm, M, order, Rn);
        }
    }
}
}  // namespace detail

// dM from the kernel values of this x point against the lane's C columns (point modes): the double increment of
// signature_algs.py:26 with the previous row kept in the lane and the left neighbour's last column handed over
template <typename T, int C, int MODE, class Lane, class Nbr>
GPSIG_HD void seq_point_increments(Lane& L, const Nbr& nbr, const T (&knew)[C], bool dummy, int rlo, int rhi, T (&dm)[C]) {
    if constexpr (MODE == MODE_PT_DIFF) {
        const T kl = nbr.kleft();              // kappa(this x point, the left neighbour's last column): it was on this row one step ago
#pragma unroll
        for (int r = 0; r < C; ++r) {
            const T e = knew[r] - (r == 0 ? kl : knew[r == 0 ? 0 : r - 1]);
            dm[r] = e - L.eprev[r];
            L.eprev[r] = e;
        }
        L.klast = knew[C - 1];
    } else {
#pragma unroll
        for (int r = 0; r < C; ++r) dm[r] = knew[r];
    }
#pragma unroll
    for (int r = 0; r < C; ++r)
        if (dummy || r < rlo || r >= rhi) dm[r] = T(0);
}

// dM for the lane's C columns from the x-side row (shared by both algorithms)
template <typename T, int C, int D, int MODE, class Lane, class Nbr>
GPSIG_HD void seq_increments(Lane& L, const Nbr& nbr, const T (&xr)[D], bool dummy, int rlo, int rhi, int kind, T p0, T p1,
                             T (&dm)[C]) {
    if constexpr (MODE == MODE_INC) {
#pragma unroll
        for (int r = 0; r < C; ++r) {
            T acc = xr[0] * L.y[r][0];
#pragma unroll
            for (int f = 1; f < D; ++f) acc = fma(xr[f], L.y[r][f], acc);
            dm[r] = acc;
        }
    } else {
        T xs = xr[0] * xr[0];
#pragma unroll
        for (int f = 1; f < D; ++f) xs = fma(xr[f], xr[f], xs);
        T knew[C];
#pragma unroll
        for (int r = 0; r < C; ++r) {
            T acc = xr[0] * L.y[r][0];
#pragma unroll
            for (int f = 1; f < D; ++f) acc = fma(xr[f], L.y[r][f], acc);
            knew[r] = acc;
        }
        base_eval_n<T, C>(kind, knew, L.y2, xs, p0, p1);
        seq_point_increments<T, C, MODE>(L, nbr, knew, dummy, rlo, rhi, dm);
    }
}

// Higher-order step (order = the reference's `order`, 2 <= order <= OMAX; order 1 also works and equals seq_step).
template <typename T, int C, int D, int MMAX, int OMAX, int MODE, class Nbr>
GPSIG_HD void seq_step(SeqLaneHO<T, C, D, MMAX, OMAX, MODE>& L, const Nbr& nbr, const T (&xr)[D], int M, int order,
                       bool dummy, int rlo, int rhi, int kind, T p0, T p1) {
    T dm[C];
    seq_increments<T, C, D, MODE>(L, nbr, xr, dummy, rlo, rhi, kind, p0, p1, dm);
    T R0[OMAX][OMAX][C];
    detail::seq_ho_level<1>(L, nbr, dm, M, order, R0);
}

template <typename T, int C, int D, int MMAX, int MODE, class Nbr>
GPSIG_HD void seq_step(SeqLane<T, C, D, MMAX, MODE>& L, const Nbr& nbr, const T (&xr)[D], int M, int /*order*/,
                       bool dummy, int rlo, int rhi, int kind, T p0, T p1) {
    T dm[C];
    seq_increments<T, C, D, MODE>(L, nbr, xr, dummy, rlo, rhi, kind, p0, p1, dm);
    seq_recursion(L, nbr, dm, M);
}

// First-order step for the RBF kernel on PRESCALED records (float64, point modes): both sides' points were multiplied by
// EXP_PRESCALE when the records were made, L.y2[r] holds -|y'_r|^2 / 2 and hx = -|x'|^2 / 2 comes with the x row (the spare
// column of the record), so  <x', y'_r> + y2[r] + hx  =  -|x - y_r|^2 / 2 * 64/ln2  is the argument of the table-driven
// 2^(t/64) (fast_exp.hpp).  Per column: D FMAs, one add, 13 instructions of exp -- against D + 3 and the ~20 of the library exp.
template <int C, int D, int MMAX, int MODE, class Nbr>
GPSIG_HD void seq_step_rbf_prescaled(SeqLane<double, C, D, MMAX, MODE>& L, const Nbr& nbr, const double (&xr)[D], double hx,
                                     const double* etab, int M, bool dummy, int rlo, int rhi) {
    static_assert(MODE != MODE_INC, "point modes only");
    double knew[C], dm[C];
#pragma unroll
    for (int r = 0; r < C; ++r) {
        double acc = fma(xr[0], L.y[r][0], L.y2[r]);
#pragma unroll
        for (int f = 1; f < D; ++f) acc = fma(xr[f], L.y[r][f], acc);
        knew[r] = SEQ_EXP256 ? kexp2_tab256(acc + hx, etab) : kexp2_tab(acc + hx, etab);
    }
    seq_point_increments<double, C, MODE>(L, nbr, knew, dummy, rlo, rhi, dm);
    seq_recursion(L, nbr, dm, M);
}

// The same for the HIGHER-ORDER algorithm (round 6: exact instances -- num_levels and order at compile time -- for the RBF kernel): the kernel values
// of the row as above, then the grid recursion of seq_ho_level with constant bounds (its `r < dcur` predicates fold away).
template <int C, int D, int MMAX, int OMAX, int MODE, class Nbr>
GPSIG_HD void seq_step_rbf_prescaled_ho(SeqLaneHO<double, C, D, MMAX, OMAX, MODE>& L, const Nbr& nbr, const double (&xr)[D], double hx,
                                        const double* etab, int M, int order, bool dummy, int rlo, int rhi) {
    static_assert(MODE != MODE_INC, "point modes only");
    double knew[C], dm[C];
#pragma unroll
    for (int r = 0; r < C; ++r) {
        double acc = fma(xr[0], L.y[r][0], L.y2[r]);
#pragma unroll
        for (int f = 1; f < D; ++f) acc = fma(xr[f], L.y[r][f], acc);
        knew[r] = SEQ_EXP256 ? kexp2_tab256(acc + hx, etab) : kexp2_tab(acc + hx, etab);
    }
    seq_point_increments<double, C, MODE>(L, nbr, knew, dummy, rlo, rhi, dm);
    double R0[OMAX][OMAX][C];
    detail::seq_ho_level<1>(L, nbr, dm, M, order, R0);
}

// First-order step for the Matern-1/2, 3/2, 5/2 kernels on PRESCALED records (float64, point modes; round 5): both sides' points were multiplied by
// S = c 256 / ln 2 (c = 1, sqrt 3, sqrt 5) when the records were made, so q = |x' - y'| = S r and exp(-c r) = 2^(-q / 256) goes through the table
// (fast_exp.hpp); u = c r = q ln2 / 256.  The squared distance is summed from the DIFFERENCES of the coordinates: exactly zero where the points
// coincide (every diagonal cell of a sequence paired with itself), where |x|^2 + |y|^2 - 2 x.y would leave rounding noise that the Matern-1/2
// kernel turns into its square root (kernels.py:765-781: the reference floors the distance at 1e-40 for the same reason).  Per column: 2 D
// instructions, an inverse square root with one Newton step, 13 of exp, 1-3 of polynomial -- against D + 3 and the ~45 of the library sqrt + exp.
constexpr bool seq_is_matern(int kind) { return kind == BASE_MATERN12 || kind == BASE_MATERN32 || kind == BASE_MATERN52; }
constexpr double seq_matern_c(int kind) { return kind == BASE_MATERN12 ? 1.0 : (kind == BASE_MATERN32 ? 1.7320508075688772935 : 2.2360679774997896964); }
constexpr double seq_matern_prescale(int kind) { return seq_matern_c(kind) * 256.0 / 0x1.62e42fefa39efp-1; }
template <int KIND, int C, int D, int MMAX, int MODE, class Nbr>
GPSIG_HD void seq_step_matern_prescaled(SeqLane<double, C, D, MMAX, MODE>& L, const Nbr& nbr, const double (&xr)[D], const double* etab, int M,
                                        bool dummy, int rlo, int rhi) {
    static_assert(MODE != MODE_INC && seq_is_matern(KIND), "point modes, Matern families");
    constexpr double S = seq_matern_prescale(KIND), K = 0x1.62e42fefa39efp-1 / 256.0, FLOOR = 1e-40 * S * S;
    double knew[C], dm[C];
#pragma unroll
    for (int r = 0; r < C; ++r) {
        double t = 0.0;
#pragma unroll
        for (int f = 0; f < D; ++f) { const double df = xr[f] - L.y[r][f]; t = fma(df, df, t); }
        const double d = t > FLOOR ? t : FLOOR;
#if defined(__HIP_DEVICE_COMPILE__)
        const double r0 = __builtin_amdgcn_rsq(d);
        double q = d * r0;
        q = fma(fma(-q, q, d), 0.5 * r0, q);            // one Newton step on the residual: ~1.5 ulp
#else
        const double q = std::sqrt(d);
#endif
        const double e = kexp2_tab256(-q, etab);
        if constexpr (KIND == BASE_MATERN12) knew[r] = e;                                               // kernels.py:955-958
        else if constexpr (KIND == BASE_MATERN32) knew[r] = fma(q, K, 1.0) * e;                        // :974-977
        else { const double u = q * K; knew[r] = fma(fma(u, 1.0 / 3.0, 1.0), u, 1.0) * e; }            // :991-993
    }
    seq_point_increments<double, C, MODE>(L, nbr, knew, dummy, rlo, rhi, dm);
    seq_recursion(L, nbr, dm, M);
}

// The same for the HIGHER-ORDER algorithm (round 6: exact instances for the Matern families as for the RBF kernel -- seq_step_rbf_prescaled_ho)
template <int KIND, int C, int D, int MMAX, int OMAX, int MODE, class Nbr>
GPSIG_HD void seq_step_matern_prescaled_ho(SeqLaneHO<double, C, D, MMAX, OMAX, MODE>& L, const Nbr& nbr, const double (&xr)[D], const double* etab, int M,
                                           int order, bool dummy, int rlo, int rhi) {
    static_assert(MODE != MODE_INC && seq_is_matern(KIND), "point modes, Matern families");
    constexpr double S = seq_matern_prescale(KIND), K = 0x1.62e42fefa39efp-1 / 256.0, FLOOR = 1e-40 * S * S;
    double knew[C], dm[C];
#pragma unroll
    for (int r = 0; r < C; ++r) {
        double t = 0.0;
#pragma unroll
        for (int f = 0; f < D; ++f) { const double df = xr[f] - L.y[r][f]; t = fma(df, df, t); }
        const double d = t > FLOOR ? t : FLOOR;
#if defined(__HIP_DEVICE_COMPILE__)
        const double r0 = __builtin_amdgcn_rsq(d);
        double q = d * r0;
        q = fma(fma(-q, q, d), 0.5 * r0, q);
#else
        const double q = std::sqrt(d);
#endif
        const double e = kexp2_tab256(-q, etab);
        if constexpr (KIND == BASE_MATERN12) knew[r] = e;
        else if constexpr (KIND == BASE_MATERN32) knew[r] = fma(q, K, 1.0) * e;
        else { const double u = q * K; knew[r] = fma(fma(u, 1.0 / 3.0, 1.0), u, 1.0) * e; }
    }
    seq_point_increments<double, C, MODE>(L, nbr, knew, dummy, rlo, rhi, dm);
    double R0[OMAX][OMAX][C];
    detail::seq_ho_level<1>(L, nbr, dm, M, order, R0);
}

// First-order step for SignatureSpectral's state-space kernel (spectral_eval above; gpsig/kernels.py:921-942), point modes: the kernel
// takes the two points themselves, which a lane has -- its C columns of y in registers, the x row of the step -- so the wavefront
// kernel carries it as a compile-time family of its own (KIND == BASE_SPECTRAL instances: no branch in anyone else's step).
// tab: alpha[Q], omega[Q][SPECTRAL_STRIDE], gamma[Q][SPECTRAL_STRIDE], zero beyond the features (the padded columns of x and y are zero too).
template <typename T, int C, int D, int MMAX, int MODE, class Nbr>
GPSIG_HD void seq_step_spectral(SeqLane<T, C, D, MMAX, MODE>& L, const Nbr& nbr, const T (&xr)[D], const double* tab, int Q, int family,
                                int M, bool dummy, int rlo, int rhi) {
    static_assert(MODE != MODE_INC, "point modes only");
    T knew[C], dm[C];
#pragma unroll
    for (int r = 0; r < C; ++r)
        knew[r] = spectral_eval<T>(tab, Q, family, D, [&](int f) { return xr[f]; }, [&](int f) { return L.y[r][f]; });
    seq_point_increments<T, C, MODE>(L, nbr, knew, dummy, rlo, rhi, dm);
    seq_recursion(L, nbr, dm, M);
}

// Per-lane position in the stream of x-side record rows, event driven: between events a step costs one add on the
// row offset and one decrement.  Events (each `period` = R1 steps apart once the lane has started): the lane's start,
// every pair boundary (row 0 of the next x: the pair that just ended is emitted, accumulators are cleared), and the
// flush step after the last x, after which the lane idles on the zero row.  Offsets are in elements from the start of
// the LDS block: [zero row: RS elements][ring: nslot slots of slot_elems].
struct LaneCtl {
    int left;     // steps until the next event (0: the event is due at this step)
    int rowoff;   // element offset of this step's record row
    int stride;   // RS while sweeping an x, 0 while idle
    int p;        // index of the current x within the task; == nx from the flush step on
    int sbase;    // element offset of the current x's ring slot
    bool row0;    // this step is row 0 of an x / the flush step (pair boundary) -- or the lane is idle

    GPSIG_HD void init(int lam, int RS) { left = lam; rowoff = 0; stride = 0; p = -1; sbase = RS; row0 = true; }
    // Call at the start of every step.  Returns true on a pair boundary (p >= 1 then names the pair p-1 that ended).
    GPSIG_HD bool begin_step(int nx, int R1, int RS, int slot_elems, int ring_elems) {
        row0 = (stride == 0);
        if (left != 0) return false;
        ++p;
        if (p > nx) { left = 0x3fffffff; return false; }          // idle for good
        if (p >= 1) { sbase += slot_elems; if (sbase == RS + ring_elems) sbase = RS; }
        row0 = true;
        if (p < nx) { rowoff = sbase; stride = RS; left = R1; }    // row 0 of x number p
        else { rowoff = 0; stride = 0; left = 1; }                 // flush step: zero row; one more event retires the lane
        return true;
    }
    GPSIG_HD void end_step() { rowoff += stride; --left; }
};

}  // namespace gpsig
