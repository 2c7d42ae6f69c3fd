// grad_fused_kernel.hpp -- reverse pass of the sequence-vs-sequence Gram for the stationary base kernels on points (SignatureRBF,
// SignatureMatern12 / 32 / 52; order 1: gpsig/kernels.py:188-237 + signature_algs.py:8-35 differentiated), without Lam ever leaving the chip
// (round 5).  Template parameters: DP padded state-space columns (4, 8; 16 with two lattice columns per lane), LQ = num_levels - 1, KIND, G lanes
// per pair (16 / 32 / 64: four / two / one pair per wavefront, up to 64 / 128 / 256 points on the column side), C lattice columns per lane,
// DIFF (the lattice of double increments, or -- difference=False -- the kernel matrix of the points itself), PHASE (0: both sweeps; 2: the
// backward sweep only, continued from the stash the evaluation kernel wrote: gpsig_seq_gram_levels_stash).
//
// A workgroup is TWO wavefronts working on the same 64 / G sequence pairs (register-side sequences r0 .. r0 + 64/G - 1, one per pair group
// of G lanes, against a run of streamed sequences s they share; lane ln of a group owns the lattice columns C ln .. C ln + C - 1):
//   wavefront 0, the EVALUATOR: keeps the lane's points of y, evaluates the kernel row of the step (table-driven exp on prescaled
//       points), hands the double increments dm of the lane's columns to the sweeper, and -- in the backward sweep -- takes back
//       W = H * g (the adjoint of the kernel values times the kernel's derivative factor) and contracts both sides: the y side into
//       per-lane accumulators that live for the whole run, the x side as a partial row sum that travels from lane to lane with the
//       skew of the sweep (one DPP shift per word and step) and leaves lane 0 into an LDS accumulator.
//   wavefront 1, the SWEEPER: the forward recursion (WaveFwd) and its undoing (WaveUndo) of grad_wave_core.hpp, dm in; Lam, its double
//       difference H and W out.
// The two exchange dm / W through same-lane LDS slots, double buffered, one workgroup barrier per step; the kernel values a
// contraction needs were evaluated three intervals earlier and wait in a five-deep same-lane ring.  Split this way each
// wavefront's state fits the 256 registers of two wavefronts per SIMD, which the one-wavefront form (seq_lam_undo_kernel:
// 256 + AGPRs, one wavefront per SIMD, Lam through HBM to lam_contract_kernel) does not.
// The interval schedule is replayed lane by lane in tools/sim_fused_grad.py against autograd of the plain recursion.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "exp_pair_asm.hpp"
#include "fast_exp.hpp"
#include "grad_wave_core.hpp"
#include "grad_wave_kernel.hpp"
#include "seq_args.hpp"

namespace gpsig {

struct FusedGradArgs {
    const double* S; const double* R;      // streamed side / register-resident side, scaled observations, user layout (N, L, d)
    double* gS; double* gR;                // their gradients, same layout, accumulated with atomics (the same array for a symmetric Gram)
    int NS, NR, LS, LR, d;
    const SeqTask* tasks;                  // y0 = first of the 64 / G register-side sequences, x0 / nx = run of streamed sequences
    const double* G; int64_t gm, gs, gr;   // upstream: G[m * gm + s * gs + r * gr], m = 1 .. M
    int sym;                               // symmetric Gram: pairs s >= r only, s > r carries G[s][r] + G[r][s]
    // PHASE == 2 (backward sweep only): the forward recursion's row totals and final Q's come from the stash the evaluation kernel wrote
    // (seq_gram_kernel.hpp: STASH; layout fused_stash_stride below) and the tasks are pieces of ITS tasks: pair k of this task, group g, sits at
    // stash + ((pair0 + k) * (64 / G) + g) * stash_stride, pair0 = (x0 << 32 | y0) of pair0_list[task]; circ: the symmetric Gram's pairs are
    // the evaluation kernel's circulant ones (SeqGramArgs: PRED_CIRCULANT -- streamed indices wrap, ownership as seq_emit decides it)
    const double* stash; const SeqTask* pair0_list; int64_t stash_stride; int circ;
};

// doubles per pair in the stash: R1 lattice rows x LQ row totals (levels 1 .. M-1), then G lanes x LQ x C final Q's (forward convention)
__host__ __device__ inline int64_t fused_stash_stride(int R1, int LQ, int G, int C) { return int64_t(R1) * LQ + int64_t(G) * LQ * C; }

constexpr int FG_KH = 5;                 // depth of the kernel-value ring.  G = 16 or 64 lanes per pair (4 pairs or 1 per wavefront), C = 4 columns per lane
                                         // (2 where the state space has 9 .. 16 columns: the lane's points are C * DP doubles of registers)

// offsets (in doubles) into the dynamic LDS of one workgroup
struct FusedLds { int etab, xs, gxa, rt, dm, lam, kh, total; };
__host__ __device__ inline FusedLds fused_lds(int LS, int R1, int DP, int LQ, int G, int C) {
    FusedLds o;
    const int DS = DP + 2;                 // record row: DP prescaled features, -|x'|^2 / 2, one pad (rows 16-byte aligned, bank-conflict free)
    int p = 0;
    o.etab = p; p += EXP_TAB256_N;
    o.xs = p; p += LS * DS;
    o.gxa = p; p += LS * DS;
    o.rt = p; p += (64 / G) * (R1 > 0 ? R1 : 1) * LQ;
    p = (p + 1) & ~1;
    o.dm = p; p += 2 * C * 64;             // [parity][half][lane] pairs of doubles
    o.lam = p; p += 2 * C * 64;
    o.kh = p; p += FG_KH * C * 64;
    o.total = p;
    return o;
}

typedef double fg_d2 __attribute__((ext_vector_type(2)));

// columns per lane for a padded feature count: the lane keeps C points of DP doubles, and as many accumulators again
constexpr int fused_grad_columns(int DP) { return DP > 8 ? 2 : 4; }

// What the kernel is built for: the stationary base kernels kappa = phi(|x - y|^2) whose exp goes through the 256-entry table.  Their gradient
// has one shape, d kappa / dx = g (x - y) with g = 2 phi', so the contraction is W = H * g whatever the family; the kernel-value ring holds -g.
//   RBF: points prescaled by sqrt(256 / ln 2), t = x'.y' - |x'|^2/2 - |y'|^2/2, kappa = 2^(t/256), -g = kappa.
//   Matern-nu: points prescaled by S = c 256 / ln 2 (c = 1, sqrt 3, sqrt 5), q = |x' - y'| = S r, e = 2^(-q/256) = exp(-c r), u = c r = q ln2/256:
//   1/2: kappa = e, -g = e / r;  3/2: kappa = (1 + u) e, -g = 3 e;  5/2: kappa = (1 + u + u^2/3) e, -g = 5/3 (1 + u) e  (kernels.py:955-993; the
//   distance floored at 1e-40 as there, where g is taken as 0 like grad_core.hpp's base_eval_grad).  The squared distance is summed from the
//   DIFFERENCES of the coordinates (2 D instructions instead of D): |x|^2 + |y|^2 - 2 x.y leaves rounding noise of 1e-16 |x|^2 where the points
//   coincide -- every diagonal cell of a sequence paired with itself -- and the Matern-1/2 kernel turns a noise eps into sqrt(eps) (kappa(x, x) =
//   1 - 3e-8, g = 1e7 instead of 1 and 0: 8e-7 on the gradient in the first build of this family).
constexpr bool fg_matern(int kind) { return kind == BASE_MATERN12 || kind == BASE_MATERN32 || kind == BASE_MATERN52; }
constexpr double fg_matern_c(int kind) { return kind == BASE_MATERN12 ? 1.0 : (kind == BASE_MATERN32 ? 1.7320508075688772935 : 2.2360679774997896964); }
constexpr double FG_LN2 = 0x1.62e42fefa39efp-1;
constexpr double fg_prescale(int kind) { return fg_matern(kind) ? fg_matern_c(kind) * 256.0 / FG_LN2 : EXP_PRESCALE256; }
constexpr double fg_row_factor(int kind) { return 1.0; }                                    // record rows hold this times the prescaled point
constexpr double fg_norm_factor(int kind) { return -0.5; }                                  // ... and this times its squared norm (RBF's argument)

// kernel values k (and -g, see above) of record row `xrow` (LDS) against the lane's four points, the exps two at a time through the
// hand-scheduled table exp (exp_pair_asm.hpp)
template <int DP, int KIND, int C>
__device__ __forceinline__ void fg_kappa_row(const double* xrow, const double (&y)[C][DP], const double (&hy)[C], unsigned tab_addr, double (&k)[C],
                                             double (&g)[C]) {
    static_assert(C % 2 == 0, "exps go in pairs");
    double x[DP], t[C];
#pragma unroll
    for (int f = 0; f < DP; f += 2) {
        const fg_d2 v = *reinterpret_cast<const fg_d2*>(xrow + f);
        x[f] = v[0]; x[f + 1] = v[1];
    }
    if constexpr (KIND == BASE_RBF) {
        const double hx = xrow[DP];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            t[c] = hx + hy[c];
#pragma unroll
            for (int f = 0; f < DP; ++f) t[c] = fma(x[f], y[c][f], t[c]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            t[c] = 0.0;
#pragma unroll
            for (int f = 0; f < DP; ++f) { const double df = x[f] - y[c][f]; t[c] = fma(df, df, t[c]); }
        }
    }
    if constexpr (KIND == BASE_RBF) {
#pragma unroll
        for (int c = 0; c < C; c += 2) {
            kexp2_pair_asm<256>(t[c], t[c + 1], tab_addr, k[c], k[c + 1]);
            __builtin_amdgcn_s_waitcnt(0xc07f);          // the block waited for its table reads: tell the compiler's counters
        }
#pragma unroll
        for (int c = 0; c < C; ++c) g[c] = k[c];
    } else {
        constexpr double S = fg_prescale(KIND), K = FG_LN2 / 256.0, FLOOR = 1e-40 * S * S;
        double q[C], yi[C], e[C];
        bool floored[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            floored[c] = !(t[c] > FLOOR);
            const double d = fmax(t[c], FLOOR);
            const double r0 = __builtin_amdgcn_rsq(d), h = 0.5 * r0;
            double r = d * r0;
            r = fma(fma(-r, r, d), h, r);                // one Newton step on the residual: ~1.5 ulp
            q[c] = r;
            if constexpr (KIND == BASE_MATERN12) yi[c] = r0 * fma(-r, r0, 2.0);      // 1 / q, one Newton step
        }
#pragma unroll
        for (int c = 0; c < C; c += 2) {
            kexp2_pair_asm<256, true>(q[c], q[c + 1], tab_addr, e[c], e[c + 1]);
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if constexpr (KIND == BASE_MATERN12) { k[c] = e[c]; g[c] = floored[c] ? 0.0 : e[c] * (yi[c] * S); }
            else if constexpr (KIND == BASE_MATERN32) { k[c] = fma(q[c], K, 1.0) * e[c]; g[c] = floored[c] ? 0.0 : 3.0 * e[c]; }
            else { const double u = q[c] * K; k[c] = fma(fma(u, 1.0 / 3.0, 1.0), u, 1.0) * e[c]; g[c] = floored[c] ? 0.0 : (5.0 / 3.0) * fma(q[c], K, 1.0) * e[c]; }
        }
    }
}

template <int C>
__device__ __forceinline__ void fg_put(double* slot, int par, int lane, const double (&v)[C]) {
    fg_d2* s = reinterpret_cast<fg_d2*>(slot) + par * (C / 2) * 64 + lane;
#pragma unroll
    for (int h = 0; h < C / 2; ++h) s[h * 64] = fg_d2{v[2 * h], v[2 * h + 1]};
}
template <int C>
__device__ __forceinline__ void fg_get(const double* slot, int par, int lane, double (&v)[C]) {
    const fg_d2* s = reinterpret_cast<const fg_d2*>(slot) + par * (C / 2) * 64 + lane;
#pragma unroll
    for (int h = 0; h < C / 2; ++h) {
        const fg_d2 a = s[h * 64];
        v[2 * h] = a[0]; v[2 * h + 1] = a[1];
    }
}

// Both wavefronts: the record of streamed sequence s -- prescaled rows and -|x'|^2 / 2 -- and a cleared x-side accumulator.
template <int DP, int KIND>
__device__ __forceinline__ void fg_stage(const FusedGradArgs& A, double* sm, const FusedLds o, int64_t s) {
    constexpr int DS = DP + 2;
    for (int p = threadIdx.x; p < A.LS; p += 128) {
        const double* src = A.S + (s * A.LS + p) * A.d;
        double* xr = sm + o.xs + p * DS;
        double* gr = sm + o.gxa + p * DS;
        double hs = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f) {
            const double v = f < A.d ? src[f] * fg_prescale(KIND) : 0.0;
            xr[f] = v * fg_row_factor(KIND);
            gr[f] = 0.0;
            hs = fma(v, v, hs);
        }
        xr[DP] = fg_norm_factor(KIND) * hs;
        xr[DP + 1] = 0.0;
        gr[DP] = gr[DP + 1] = 0.0;
    }
}
// Both wavefronts, after the last backward interval: gx[p] = (sum_q W[p][q]) x_p - sum_q W[p][q] y_q, back in the caller's scale
// (the record rows hold fg_row_factor times the prescaled point, the accumulated sum the prescaled y's).
template <int DP, int KIND>
__device__ __forceinline__ void fg_flush(const FusedGradArgs& A, const double* sm, const FusedLds o, int64_t s) {
    constexpr int DS = DP + 2;
    constexpr double ca = 1.0 / (fg_prescale(KIND) * fg_row_factor(KIND)), cb = -1.0 / fg_prescale(KIND);
    for (int e = threadIdx.x; e < A.LS * DP; e += 128) {
        const int p = e / DP, f = e % DP;
        if (f < A.d) {
            const double* xr = sm + o.xs + p * DS;
            const double* gr = sm + o.gxa + p * DS;
            atomicAdd(&A.gS[(s * A.LS + p) * A.d + f], fma(gr[DP] * ca, xr[f], cb * gr[f]));
        }
    }
}

// ---- wavefront 0 ----------------------------------------------------------------------------------------------------------------
template <int DP, int KIND, int G, int C, bool DIFF, int PHASE>
__device__ __forceinline__ void fg_evaluator(const FusedGradArgs& A, const SeqTask tk, double* sm, const FusedLds o, int R1, int R2, int TF) {
    constexpr int DS = DP + 2;
    const int lane = threadIdx.x & 63, ln = lane & (G - 1), gw = lane / G;
    const double* xs = sm + o.xs;
    double* gxa = sm + o.gxa;
    const int64_t r = int64_t(tk.y0) + gw;
    const bool rvalid = r < A.NR;
    const int b0 = C * ln;
    const unsigned tab_addr = __builtin_amdgcn_readfirstlane(unsigned(uintptr_t((__attribute__((address_space(3))) const void*)(sm + o.etab))));
    int nvalid = R2 - b0;                     // lattice columns among the lane's (difference=False: its points inside the sequence)
    nvalid = nvalid < 0 ? 0 : (nvalid > C ? C : nvalid);

    // The lane's points b0 .. b0 + 3.  Beyond the sequence the LAST point repeats: the columns there get dm == 0 exactly (equal
    // arguments, equal kernel values) without a mask in the evaluation.
    double y[C][DP], hy[C], ay[C][DP], by[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int q = b0 + c < A.LR ? b0 + c : A.LR - 1;
        const double* src = A.R + ((rvalid ? r : 0) * A.LR + q) * A.d;
        double s = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f) {
            const double v = (rvalid && f < A.d) ? src[f] * fg_prescale(KIND) : 0.0;
            y[c][f] = v;
            s = fma(v, v, s);
        }
        hy[c] = fg_norm_factor(KIND) * s;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        by[c] = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f) ay[c][f] = 0.0;
    }
    const int lsm1 = A.LS - 1;
    auto row_of = [&](int p) { return xs + (p < 0 ? 0 : (p > lsm1 ? lsm1 : p)) * DS; };

    for (int it = 0; it < tk.nx; ++it) {
        int64_t s = int64_t(tk.x0) + it;
        if (PHASE == 2 && A.circ && s >= A.NS) s -= A.NS;
        __syncthreads();                       // the flush of the previous streamed sequence has read gxa / xs
        fg_stage<DP, KIND>(A, sm, o, s);
        __syncthreads();
        // ---- forward sweep: dm of step i in interval i.  (difference=False, DIFF == false: the lattice is the kernel matrix of the points itself --
        // the lane's columns are its points in both sweeps, dm = kappa, forced to zero beyond the sequence.)  With differences: here the lane's four columns are b0-1 .. b0+2 -- the differences that END at its own
        // points, the kernel value at b0-1 being the left neighbour's last, one interval old (it runs one row ahead): four evaluations per
        // row, the evaluation kernel's convention (seq_core.hpp).  Lane 0's column -1 is no column: dm == 0 there.  A lane ahead of its first
        // row evaluates row 0 again and again (row_of clamps), so rd needs no guard; what it hands over outside its rows the sweeper does not read.
        double rd[C], k3 = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) rd[c] = 0.0;
        if constexpr (DIFF && PHASE != 2) {
            double k[C], g[C];
            fg_kappa_row<DP, KIND, C>(row_of(0), y, hy, tab_addr, k, g);
            double kl = wave_from_left<G>(k[C - 1]);
            if (ln == 0) kl = k[0];
            rd[0] = k[0] - kl;
#pragma unroll
            for (int c = 1; c < C; ++c) rd[c] = k[c] - k[c - 1];
            k3 = k[C - 1];
        }
        for (int i = 0; i <= (PHASE == 2 ? -1 : TF); ++i) {     // (PHASE == 2: no forward sweep, its results are in the stash)
            if (i < TF) {
                double k[C], g[C], dm[C];
                if constexpr (DIFF) {
                    double kl = wave_from_left<G>(k3);
                    fg_kappa_row<DP, KIND, C>(row_of(i - ln + 1), y, hy, tab_addr, k, g);
                    if (ln == 0) kl = k[0];
                    k3 = k[C - 1];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const double nd = k[c] - (c == 0 ? kl : k[c > 0 ? c - 1 : 0]);
                        dm[c] = nd - rd[c];
                        rd[c] = nd;
                    }
                } else {
                    fg_kappa_row<DP, KIND, C>(row_of(i - ln), y, hy, tab_addr, k, g);
#pragma unroll
                    for (int c = 0; c < C; ++c) dm[c] = c < nvalid ? k[c] : 0.0;
                }
                fg_put(sm + o.dm, i & 1, lane, dm);
            }
            __syncthreads();
        }
        // ---- backward sweep.  The lane's columns are now b0 .. b0+3, the differences that START at its points.  Interval i: the kernel row
        // a = R1 + (G-1-ln) - i (a == R1 primes rd; beyond it row R1 again); the value at b0+4 is the right neighbour's first, one interval
        // old (it runs one row ahead in this direction); the last lane's column 63 is never a lattice column (at most 64 points).  The
        // sweeper takes dm in interval i + 1 and hands back W = -H * kappa (the adjoint of the kernel values times the kernel's
        // derivative, formed there) of point row p = a + 4 in interval i + 2; it is contracted here in interval i + 3.
        double k0 = 0.0, P[DP + 1];
#pragma unroll
        for (int f = 0; f <= DP; ++f) P[f] = 0.0;
        int wr = 0;                            // i % FG_KH
        for (int i = 0; i <= TF + 4; ++i) {
            if (i <= TF) {
                double k[C], g[C], dm[C];
                if constexpr (DIFF) {
                    double kr = wave_from_right<G>(k0);
                    fg_kappa_row<DP, KIND, C>(row_of(R1 + (G - 1 - ln) - i), y, hy, tab_addr, k, g);
                    if (ln == G - 1) kr = k[C - 1];
                    k0 = k[0];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const double nd = (c == C - 1 ? kr : k[c < C - 1 ? c + 1 : c]) - k[c];
                        dm[c] = rd[c] - nd;
                        rd[c] = nd;
                    }
                } else {
                    fg_kappa_row<DP, KIND, C>(row_of(R1 + (G - 1 - ln) - i), y, hy, tab_addr, k, g);
#pragma unroll
                    for (int c = 0; c < C; ++c) dm[c] = c < nvalid ? k[c] : 0.0;
                }
                fg_put(sm + o.dm, i & 1, lane, dm);
                fg_put(sm + o.kh, wr, lane, g);
            }
            if (i >= 3) {
                const int p = R1 + (DIFF ? 4 : 2) + (G - 1 - ln) - i;      // the point row of the W handed over in the previous interval
                double w[C];
                fg_get(sm + o.lam, (i - 1) & 1, lane, w);           // zeros outside the lattice's point rows
                // y side: the lane's own points
                {
                    const double* xr = row_of(p);
                    double x[DP];
#pragma unroll
                    for (int f = 0; f < DP; f += 2) {
                        const fg_d2 v = *reinterpret_cast<const fg_d2*>(xr + f);
                        x[f] = v[0]; x[f + 1] = v[1];
                    }
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        by[c] += w[c];
#pragma unroll
                        for (int f = 0; f < DP; ++f) ay[c][f] = fma(w[c], x[f], ay[c][f]);
                    }
                }
                // x side: the row sum over the group's points, handed from right to left with the skew of the sweep
#pragma unroll
                for (int f = 0; f <= DP; ++f) P[f] = wave_from_right<G>(P[f]);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    P[DP] += w[c];
#pragma unroll
                    for (int f = 0; f < DP; ++f) P[f] = fma(w[c], y[c][f], P[f]);
                }
                if (ln == 0 && p >= 0 && p <= lsm1) {
                    double* gr = gxa + p * DS;
#pragma unroll
                    for (int f = 0; f <= DP; ++f) atomicAdd(gr + f, P[f]);
                }
            }
            wr = wr + 1 == FG_KH ? 0 : wr + 1;
            __syncthreads();
        }
        fg_flush<DP, KIND>(A, sm, o, s);
    }
    if (rvalid) {
        // gy[q] = (sum_p W[p][q]) y_q - sum_p W[p][q] x_p: ay holds the record rows' multiple of x
        constexpr double ca = 1.0 / fg_prescale(KIND), cb = -1.0 / (fg_prescale(KIND) * fg_row_factor(KIND));
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int q = b0 + c;
            if (q < A.LR) {
#pragma unroll
                for (int f = 0; f < DP; ++f)
                    if (f < A.d) atomicAdd(&A.gR[(r * A.LR + q) * A.d + f], fma(by[c] * ca, y[c][f], cb * ay[c][f]));
            }
        }
    }
}

// ---- wavefront 1 ----------------------------------------------------------------------------------------------------------------
template <int DP, int LQ, int KIND, int G, int C, bool DIFF, int PHASE>
__device__ __forceinline__ void fg_sweeper(const FusedGradArgs& A, const SeqTask tk, double* sm, const FusedLds o, int R1, int R2, int TF) {
    constexpr int M = LQ + 1;
    const int lane = threadIdx.x & 63, ln = lane & (G - 1), gw = lane / G;
    double* rt = sm + o.rt + gw * (R1 > 0 ? R1 : 1) * LQ;
    const int64_t r = int64_t(tk.y0) + gw;
    const bool rvalid = r < A.NR;
    int nvalid = R2 - C * ln;                 // lattice columns among b0 .. b0+3 (the backward sweep's)
    nvalid = nvalid < 0 ? 0 : (nvalid > C ? C : nvalid);

    int64_t pair0 = 0;
    if constexpr (PHASE == 2) {
        const SeqTask pz = A.pair0_list[blockIdx.x];
        pair0 = (int64_t(pz.x0) << 32) | int64_t(uint32_t(pz.y0));
    }
    for (int it = 0; it < tk.nx; ++it) {
        int64_t s = int64_t(tk.x0) + it;
        if (PHASE == 2 && A.circ && s >= A.NS) s -= A.NS;
        bool have = rvalid && (!A.sym || s >= r);
        if (PHASE == 2 && A.circ) {            // the evaluation kernel's ownership of the symmetric Gram's pairs (seq_emit, PRED_CIRCULANT)
            const int64_t N = A.NS, H = N / 2;
            int64_t dlt = r - s;
            if (dlt < 0) dlt += N;
            have = rvalid && (dlt < H || (dlt == H && ((N & 1) || s < r)));
        }
        double clev[LQ + 2];
#pragma unroll
        for (int p = 0; p < LQ + 2; ++p) {
            double v = 0.0;
            if (have && p >= 1 && p <= M) {
                v = A.G[p * A.gm + s * A.gs + r * A.gr];
                if (A.sym && s != r) v += A.G[p * A.gm + r * A.gs + s * A.gr];
            }
            clev[p] = v;
        }
        __syncthreads();
        fg_stage<DP, KIND>(A, sm, o, s);
        __syncthreads();
        WaveFwd<C, LQ> fw;
        fw.reset();
        if constexpr (PHASE == 2) {            // the forward sweep's results from the stash: row totals into LDS, this lane's Q's into registers
            const double* st = A.stash + ((pair0 + it) * (64 / G) + gw) * A.stash_stride;
            if (rvalid) {
                // (R1 <= 63 rows here: at most 16 words per lane -- all loads in flight before the first LDS store)
                double v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) { const int e = ln + k * G; v[k] = e < R1 * LQ ? st[e] : 0.0; }
#pragma unroll
                for (int k = 0; k < 16; ++k) { const int e = ln + k * G; if (e < R1 * LQ) rt[e] = v[k]; }
                for (int e = ln + 16 * G; e < R1 * LQ; e += G) rt[e] = st[e];
                const double* qs = st + int64_t(R1) * LQ + ln * (LQ * C);
#pragma unroll
                for (int m = 0; m < LQ; ++m)
#pragma unroll
                    for (int c = 0; c < C; ++c) fw.q[m][c] = qs[m * C + c];
            } else {
                for (int e = ln; e < R1 * LQ; e += G) rt[e] = 0.0;
            }
        }
        for (int i = 0; i <= (PHASE == 2 ? -1 : TF); ++i) {
            if (i >= 1) {
                double cin[LQ + 2];
                cin[0] = 0.0;
#pragma unroll
                for (int m = 1; m < LQ + 2; ++m) cin[m] = wave_from_left<G>(fw.sout[m]);
                const int a = (i - 1) - ln;
                if (a >= 0 && a < R1) {
                    double dm[C];
                    fg_get(sm + o.dm, (i - 1) & 1, lane, dm);
                    fw.step(dm, cin, M);
                    if (ln == G - 1) {                 // dm == 0 beyond the sequence: the last lane's row prefix is the row total whatever R2
#pragma unroll
                        for (int m = 1; m <= LQ; ++m) rt[a * LQ + m - 1] = fw.sout[m];
                    }
                }
            }
            __syncthreads();
        }
        // The forward sweep's columns were b0-1 .. b0+2 (the evaluator's forward convention), the backward sweep's are b0 .. b0+3: its state
        // moves one column to the left, the last one arriving from the right neighbour.  The upstream gradients ride in the suffix sums from
        // the start (U_p = c_p + Qb_p: one add per cell and level less) -- step() is handed -0.0 in their place, which the compiler folds away.
        // Row 0's D is not forced to zero (first_row = false): the residue of the undoing there is that of any other row.
        WaveUndo<C, LQ> bw;
        double clevz[LQ + 2];
#pragma unroll
        for (int p = 0; p < LQ + 2; ++p) clevz[p] = p == M ? clev[p] : -0.0;
#pragma unroll
        for (int m = 0; m < LQ; ++m) {
            if constexpr (DIFF) {
                bw.qfg[m] = fw.q[m][0];
#pragma unroll
                for (int c = 0; c + 1 < C; ++c) bw.qf[m][c] = fw.q[m][c + 1];
                bw.qf[m][C - 1] = wave_from_right<G>(fw.q[m][0]);
            } else {                                // difference=False: the same columns in both sweeps
                bw.qfg[m] = fw.qg[m];
#pragma unroll
                for (int c = 0; c < C; ++c) bw.qf[m][c] = fw.q[m][c];
            }
            bw.qbg[m] = clev[m + 1];
#pragma unroll
            for (int c = 0; c < C; ++c) bw.qb[m][c] = clev[m + 1];
            bw.svout[m] = bw.sufout[m] = 0.0;
        }
        // Interval i: step i - 2 of the undoing sweep (row a of the lane), then the adjoint of the kernel values from Lam:
        //   E[p][b] = Lam[p-1][b] - Lam[p][b]  (formed for p = a + 1 as the rows come, Lam == 0 outside the lattice),
        //   H[p][q] = E[p][q-1] - E[p][q]      (for p = a + 2: the left neighbour's last column is one interval behind),
        //   W = H * g(x_p, y_q)                (-g from the evaluator's ring, evaluated in interval i - 3; RBF: g = -kappa),
        // handed to the evaluator for every lane and interval it reads.  Outside the point rows 0 .. R1 H is zero by construction and the
        // ring holds finite values (cleared at the start of the task), so W is zero there without a guard.
        double lamk[C], ep[C];
#pragma unroll
        for (int c = 0; c < C; ++c) lamk[c] = ep[c] = 0.0;
        int rdk = FG_KH - 3;                   // (i - 3) % FG_KH
        for (int i = 0; i <= TF + 4; ++i) {
            if (i >= 2 && i <= TF + 3) {
                double sufin[LQ], svin[LQ], lv[C];
#pragma unroll
                for (int p = 0; p < LQ; ++p) {
                    sufin[p] = wave_from_right<G>(bw.sufout[p]);
                    svin[p] = wave_from_right<G>(bw.svout[p]);
                }
                const int a = R1 - 1 - ((i - 2) - (G - 1 - ln));
                const bool act = a >= 0 && a < R1;
#pragma unroll
                for (int c = 0; c < C; ++c) lv[c] = 0.0;
                if (act) {
                    double dm[C], rtv[LQ];
                    fg_get(sm + o.dm, (i - 1) & 1, lane, dm);
#pragma unroll
                    for (int p = 0; p < LQ; ++p) rtv[p] = rt[a * LQ + p];
                    bw.step(dm, clevz, rtv, sufin, svin, M, false, ln == 0, lv);
                }
                double w[C], kp[C];
                if constexpr (DIFF) {
                    double en[C], h[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const double li = c < nvalid ? lv[c] : 0.0;
                        en[c] = li - lamk[c];
                        lamk[c] = li;
                    }
                    const double eleft = wave_from_left<G>(en[C - 1]);
                    h[0] = eleft - ep[0];
#pragma unroll
                    for (int c = 1; c < C; ++c) h[c] = ep[c - 1] - ep[c];
#pragma unroll
                    for (int c = 0; c < C; ++c) ep[c] = en[c];
                    fg_get(sm + o.kh, rdk, lane, kp);
#pragma unroll
                    for (int c = 0; c < C; ++c) w[c] = -(h[c] * kp[c]);
                } else {                            // difference=False: H = Lam; this row's kernel values were evaluated in the previous interval
                    fg_get(sm + o.kh, rdk + 2 >= FG_KH ? rdk + 2 - FG_KH : rdk + 2, lane, kp);
#pragma unroll
                    for (int c = 0; c < C; ++c) w[c] = c < nvalid ? -(lv[c] * kp[c]) : 0.0;
                }
                fg_put(sm + o.lam, i & 1, lane, w);
            }
            rdk = rdk + 1 == FG_KH ? 0 : rdk + 1;
            __syncthreads();
        }
        fg_flush<DP, KIND>(A, sm, o, s);
    }
}

// grid: one workgroup of 128 threads per task; dynamic LDS: fused_lds(LS, LS - 1, DP, LQ).total doubles
template <int DP, int LQ, int KIND, int G, int C, bool DIFF, int PHASE = 0>
__global__ void __launch_bounds__(128, 2) seq_grad_fused_kernel(const FusedGradArgs A) {
    extern __shared__ __attribute__((aligned(16))) double fg_sm[];
    static_assert(G == 16 || G == 32 || G == 64, "a pair group is a DPP row, half the wavefront or all of it");
    const int R1 = A.LS - (DIFF ? 1 : 0), R2 = A.LR - (DIFF ? 1 : 0), TF = R1 + G - 1;
    const FusedLds o = fused_lds(A.LS, R1, DP, LQ, G, C);
    const SeqTask tk = A.tasks[blockIdx.x];
    const int role = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
    exp_tab256_fill(fg_sm + o.etab, int(threadIdx.x), 128);
    for (int e = threadIdx.x; e < FG_KH * C * 64; e += 128) fg_sm[o.kh + e] = 0.0;
    // Both wavefronts run the same barrier sequence: per streamed sequence two around the staging of its record (fg_stage), then
    // TF + 1 forward and TF + 5 backward intervals; the flush of the x side (fg_flush) is covered by the next sequence's first barrier.
#if defined(FG_ONLY_ROLE)          // register count of one role alone (compile-time experiment; such a kernel deadlocks at its first barrier)
    if (role == FG_ONLY_ROLE) { if (FG_ONLY_ROLE == 0) fg_evaluator<DP, KIND, G, C, DIFF, PHASE>(A, tk, fg_sm, o, R1, R2, TF); else fg_sweeper<DP, LQ, KIND, G, C, DIFF, PHASE>(A, tk, fg_sm, o, R1, R2, TF); }
#else
    if (role == 0) fg_evaluator<DP, KIND, G, C, DIFF, PHASE>(A, tk, fg_sm, o, R1, R2, TF);
    else fg_sweeper<DP, LQ, KIND, G, C, DIFF, PHASE>(A, tk, fg_sm, o, R1, R2, TF);
#endif
}

}  // namespace gpsig
