// tvs_tile_kernel.hpp -- inducing tensors vs sequences (Kzx) for many tensors: SignatureKernel._K_tens_vs_seq +
// signature_kern_tens_vs_seq_first_order (gpsig/kernels.py:313-340, gpsig/signature_algs.py:101-127) + the epilogue of
// K_tens_vs_seq (kernels.py:572-588), order 1.
//
// Mapping (round 5).  A WAVEFRONT owns a block of 64 inducing tensors (lane = tensor) and a run of consecutive sequences, which it draws
// from a queue (persistent launch: as many workgroups as the chip holds, long runs first, short ones last).  The levels are split into P sets
// (a level's chain only involves its own components, signature_algs.py:118-125; longest-processing-time assignment, tvs_plan.hpp) and the wave
// sweeps a tile of 16 sequences once per set, with that set's components in registers: M = 4 with increments is three sets of 4 / 3 / 3
// components x 2 points x d doubles, which fits the 168 registers of THREE wavefronts per SIMD where all ten components would not fit one.
// (Rounds 2-4 gave the sets to the NW waves of a workgroup, which then waited for each other at a barrier per sequence -- the heaviest set has a
// third more exps than the others -- and staged the sequence through LDS.)  A sequence's rows are wave-uniform, so they arrive by SCALAR loads
// (one row ahead of the sweep) and feed the inner products as scalar operands: no LDS traffic, no vector registers.  One sweep over time per
// sequence and set with the running sums of the chains in registers,
//     u[k0+j] += dM[k0+j](tau) * u[k0+j-1]   (j = i-1 .. 1, old values),   u[k0] += dM[k0](tau),   K_i = u[k0+i-1],
// no cross-lane traffic.  Results collect in the wave's own LDS tile [tensor][sequence], the sets adding up in a fixed order, and leave 16
// sequences at a time, so that every store instruction writes four full 128-byte lines of the (T, N) result.  The four waves of a workgroup
// share nothing but the exp table.
//
// Base kernel at compile time: KIND = BASE_LINEAR (records hold increments of the scaled sequence when difference is on,
// tensors with increments are collapsed to z1 - z0 on the way in: one inner product per component and time step),
// KIND = BASE_RBF (points prepared in units of sqrt(ln2/256), so the inner products are the argument of the table-driven
// 2^(t/256), fast_exp.hpp), or KIND = -1: the family is a run-time value (base_eval_n of seq_core.hpp).
#pragma once

#include "aux_kernels.hpp"
#include "fast_exp.hpp"
#include "exp_pair_asm.hpp"
#include "tvs_plan.hpp"

namespace gpsig {

// Entries of the exp table (fast_exp.hpp): 64 (degree-5 tail), 256 (degree 4), 1024 / 2048 (degree 3, one instruction fewer per exp); 32 = the
// two-level form of the 1024-entry table (two bank-conflict-free tables of 32 entries, one more read and multiplication per exp).
// Same box, alternating processes: 64 -> 256 entries: Kzx RBF 3.20 -> 3.07 ms, with increments 6.49 -> 6.17 ms (profiles/r02_ab_exp256.txt);
// 256 -> 1024 -> 2048 entries: 2.93 -> 3.34 -> 3.70 ms WITHOUT increments (8 / 16 KB more LDS per workgroup cost the stream-bound case
// its occupancy), 6.07 -> 5.88 -> 5.89 ms WITH increments (ten exps per step and wave: the instruction counts) -- profiles/r02_ab_exptab.txt.
// Hence by kernel: incremental tensors take 1024 entries, the others 256.  TVS_EXPTAB (32, 64 ... 2048) forces one size for A/B builds.
#ifdef TVS_EXPTAB
constexpr int tvs_etab_n(bool) { return TVS_EXPTAB; }
#else
constexpr int tvs_etab_n(bool two_points) { return two_points ? 1024 : 256; }
#endif
constexpr int tvs_etab_doubles(bool two_points) { return tvs_etab_n(two_points) == 32 ? EXP_TAB2L_N : tvs_etab_n(two_points); }
constexpr double tvs_rbf_prescale(bool two_points) {
    return tvs_etab_n(two_points) == 64 ? EXP_PRESCALE : (tvs_etab_n(two_points) == 256 ? EXP_PRESCALE256 :
           ((tvs_etab_n(two_points) == 1024 || tvs_etab_n(two_points) == 32) ? 4.0 * EXP_PRESCALE : 0x1.b2da4e9808a53p+5));
}
template <int NTAB>
__device__ __forceinline__ double tvs_exp2(double t, const double* etab) {
    if constexpr (NTAB == 32) return kexp2_tab2l(t, etab);
    else if constexpr (NTAB == 64) return kexp2_tab(t, etab);
    else if constexpr (NTAB == 256) return kexp2_tab256(t, etab);
    else return kexp2_tabn<NTAB>(t, etab);
}
// Matern families at compile time (round 5).  With r = sqrt(max(|x - z|^2, 1e-40)) (kernels.py:779-781) the kernels are f(c r) exp(-c r), c = 1,
// sqrt 3, sqrt 5 (:955-993).  Points prepared in units of 1/s, s = c N / ln2 (N the exp table's entries), the tensors' components times -2 on top:
// then  q^2 = (|z'|^2 + |x'|^2) + sum_f (-2 z'_f) x'_f  costs what the RBF argument costs, q = s r IS minus the argument of the table-driven
// 2^(t/N), and u = c r = q ln2 / N feeds the polynomial factor.  sqrt: v_rsq_f64 and one Newton step on the residual (5 instructions).
constexpr bool tvs_is_matern(int kind) { return kind == BASE_MATERN12 || kind == BASE_MATERN32 || kind == BASE_MATERN52; }
constexpr double tvs_matern_c(int kind) { return kind == BASE_MATERN12 ? 1.0 : (kind == BASE_MATERN32 ? 1.7320508075688772935 : 2.2360679774997896964); }
constexpr double TVS_LN2 = 0x1.62e42fefa39efp-1;
constexpr double tvs_matern_prescale(int kind, bool two_points) { return tvs_matern_c(kind) * double(tvs_etab_n(two_points) == 32 ? 1024 : tvs_etab_n(two_points)) / TVS_LN2; }
constexpr int TVS_TILE_S = 16;           // sequences per output flush: 16 doubles = one 128-byte line per tensor row
constexpr int TVS_REC_ALIGN = 128;       // (records of the reverse pass, tvs_grad_tile_kernel.hpp: granule of its LDS-DMA staging)
// A sequence's record is L rows of RS = D + 1 doubles: the D prepared features and the squared norm of the row's POINT, read by scalar loads
constexpr int tvs_row_stride(int D) { return D + 1; }
using tvs_cptr = const __attribute__((address_space(4))) double*;       // read-only rows through the scalar data cache

struct TvsTileArgs {
    const void* XR;     // (N * L + 1, RS) rows: D prepared values (D = the kernel's feature width, zero beyond d), then the squared norm of the
                        // row's POINT (linear kernel with differences: rows are increments, row 0 of a sequence zero)
    const void* ZL;     // (lt, E, D, Tpad) prepared tensor components, tensor index fastest, zero rows beyond d
    const void* ZN;     // (lt, E, Tpad) squared norms of the prepared components
    int64_t N, Tn, Tpad;
    int32_t L, d, kind, difference, M;
    int32_t run;        // the launch is persistent: workgroups draw (tensor block, run of sequences) items from queue[tensor block]; item i of a
    int32_t cnt1;       // block covers `run` sequences for i < cnt1, then `run2` for the next cnt2 items, then `run3` (TvsTileArgs::item): long
    int32_t run2, cnt2; // runs first, short ones at the end, so that the workgroups run dry together
    int32_t run3;
    int32_t items;      // items per tensor block
    int32_t* queue;     // (Tpad / 64) counters, zero at launch (prep_tensors_tile_kernel clears them)
    double p0, p1;
    const void* fx;     // (N, M+1) per-sequence factors or NULL
    const double* w;    // (M+1) level weights or NULL
    void* out;          // (T, N) or (M+1, T, N)
    int32_t sum_levels;
    int32_t order;      // higher-order instances (HO): the reference's `order` >= 2 (signature_algs.py:129-160); first-order ones ignore it
    double* aux;        // optional (N, lt, Tpad): the totals of EVERY chain, u_{j+1} of component k = i(i-1)/2 + j at [n][k][t] -- what the
                        // reverse pass (tvs_grad_tile_kernel.hpp) would otherwise rebuild with a forward sweep of its own
    // first sequence and length of item i of a tensor block
    __host__ __device__ void item(int i, int64_t* begin, int64_t* end) const {
        int64_t b, len;
        if (i < cnt1) { b = int64_t(i) * run; len = run; }
        else if (i < cnt1 + cnt2) { b = int64_t(cnt1) * run + int64_t(i - cnt1) * run2; len = run2; }
        else { b = int64_t(cnt1) * run + int64_t(cnt2) * run2 + int64_t(i - cnt1 - cnt2) * run3; len = run3; }
        *begin = b;
        *end = b + len < N ? b + len : N;
    }
    // the item schedule for `workers` wavefronts sharing one tensor block: 70 % of the sequences in runs of r, 20 % in runs of r/2, the rest in
    // runs of r/8.  A worker's share is small (configs[2]: 16,384 sequences over 256 workers per tensor block = 64), so the last items decide how
    // evenly the launch ends: runs of (8, 4, 2) against (16, 8, 4) at a share of 43 were 5.44 against 5.80 ms (profiles/r05_ab_c3.txt)
    void plan_items(int64_t workers) {
        const int64_t share = (N + workers - 1) / (workers < 1 ? 1 : workers);
        int r = share >= 512 ? 32 : (share >= 128 ? 16 : 8);
        run = r; run2 = r / 2; run3 = r / 8;
        cnt1 = int32_t((N * 7 / 10) / run);
        cnt2 = int32_t((N * 2 / 10) / run2);
        const int64_t rest = N - int64_t(cnt1) * run - int64_t(cnt2) * run2;
        items = cnt1 + cnt2 + int32_t((rest + run3 - 1) / run3);
    }
};

constexpr int TVS_WG_WAVES = 4;          // independent wavefronts per workgroup (one per SIMD), sharing the exp table
// tile slots of one wave: the level sum, or -- level arrays requested -- the levels of the largest set and level 0 (a set is flushed before the next)
constexpr int tvs_set_levels(int mask) {
    int n = 0;
    for (int i = 1; i < 16; ++i) n += (mask >> i) & 1;
    return n;
}
constexpr int tvs_tile_slots(int M, int P, bool sum_levels) {
    if (sum_levels) return 1;
    int b = 0;
    for (int p = 0; p < P; ++p) {
        const int c = tvs_set_levels(tvs_level_mask(M, P, p)) + (p == 0 ? 1 : 0);
        if (c > b) b = c;
    }
    return b;
}
// LDS bytes of one workgroup: the exp table and the waves' result tiles
inline size_t tvs_tile_lds_bytes(int M, int P, bool sum_levels, bool two_points) {
    return sizeof(double) * (tvs_etab_doubles(two_points) + size_t(TVS_WG_WAVES) * tvs_tile_slots(M, P, sum_levels) * 64 * (TVS_TILE_S + 1));
}

template <int N, bool NEG = false>
__device__ __forceinline__ void tvs_exp2_pair(double t0, double t1, unsigned tab, double& e0, double& e1) { kexp2_pair_asm<N, NEG>(t0, t1, tab, e0, e1); }

// The families the run-time-family instance (KIND == -1) still serves -- cosine, poly, mix; linear, RBF and the Matern families have instances of
// their own -- evaluated like base_eval_n of seq_core.hpp, whose switch over all nine families (library pow, exp and sqrt per value, unrolled)
// made these instances 100-200 KB each.
template <int NV>
__device__ __forceinline__ void tvs_base_eval_rest(int kind, double (&v)[NV], const double (&a2)[NV], double b2, double p0, double p1) {
    if (kind == BASE_COSINE) {
        const double sb = sqrt(b2);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = v[i] / (sb * sqrt(a2[i]));
    } else if (kind == BASE_POLY) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = poly_pow(v[i] + p0, p1);
    } else if (kind == BASE_MIX) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = p0 * kexp(-fma(-2.0, v[i], b2 + a2[i]) / 2) + (1.0 - p0) * v[i];
    } else {
        __builtin_trap();                                          // the host's dispatch sends no other family here (api.hip, tens_vs_seq_tile_device)
    }
}

template <int D>
struct TvsRow {           // one row of a record: wave-uniform, i.e. scalar registers
    double x[D];
    double xs;            // squared norm of the row's point
};
// D + 1 scalar loads of one double each.  (A row padded to 64 bytes and read by ONE s_load_dwordx16 leaves dead scalar registers inside the load's
// destination; the compiler reuses them at once and then has to wait for the load right behind its issue -- the whole scalar latency exposed in
// every step.  Rows without padding cannot be read wider than their 8-byte alignment, so nothing dead is ever loaded.)
template <int D>
__device__ __forceinline__ TvsRow<D> tvs_load_row(tvs_cptr rows, int64_t g) {
    constexpr int RS = tvs_row_stride(D);
    tvs_cptr p = rows + g * RS;
    TvsRow<D> r;
#pragma unroll
    for (int k = 0; k < D; ++k) r.x[k] = p[k];
    r.xs = p[D];
    return r;
}

template <int M, int P, int D, bool INCR, int KIND, int MASK, bool HO = false>
struct TvsTileWave {
    static constexpr int E = (INCR && KIND != BASE_LINEAR) ? 2 : 1;        // linear + increments arrives collapsed
    static constexpr int NC = tvs_mask_comps(MASK);
    using Row = TvsRow<D>;

    double z[NC][E][D];
    double zn[NC][E];       // RBF: -|z|^2/2 ; otherwise |z|^2
    double u[NC];

    __device__ __forceinline__ void load(const TvsTileArgs& A, int64_t t) {
        const double* __restrict__ ZL = static_cast<const double*>(A.ZL);
        const double* __restrict__ ZN = static_cast<const double*>(A.ZN);
#pragma unroll
        for (int i = 1; i <= M; ++i) {
            if (!((MASK >> i) & 1)) continue;
#pragma unroll
            for (int j = 0; j < i; ++j) {
                const int c = tvs_local_off(MASK, i) + j, k = i * (i - 1) / 2 + j;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const double s = ZN[(int64_t(k) * E + e) * A.Tpad + t];
                    zn[c][e] = KIND == BASE_RBF ? -0.5 * s : (tvs_is_matern(KIND) ? 0.25 * s : s);     // (Matern: the components carry a factor -2)
#pragma unroll
                    for (int f = 0; f < D; ++f)
                        z[c][e][f] = ZL[((int64_t(k) * E + e) * D + f) * A.Tpad + t];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) u[c] = 0.0;
    }

    // k1[c]: the component kernels at the row's time (kernels.py:323-330), before the difference along time
    __device__ __forceinline__ void eval(const TvsTileArgs& A, const Row& row, const double* __restrict__ etab, double (&k1)[NC]) const {
        double kv[NC * E], a2[NC * E];
        if constexpr (KIND == BASE_RBF) {
            const double hx = -0.5 * row.xs;
            constexpr int NTAB = tvs_etab_n(E == 2);
            if constexpr (NTAB == 32) {                           // (A/B form: two conflict-free tables, plain reads)
#pragma unroll
                for (int p = 0; p < NC * E; ++p) {
                    double t = zn[p / E][p % E] + hx;
#pragma unroll
                    for (int f = 0; f < D; ++f) t = fma(z[p / E][p % E][f], row.x[f], t);
                    kv[p] = tvs_exp2<NTAB>(t, etab);
                }
            } else {
                // two exps per block of hand-scheduled instructions (tvs_exp2_pair); an odd last argument goes through the plain routine
                const unsigned tab_addr = __builtin_amdgcn_readfirstlane(unsigned(uintptr_t((__attribute__((address_space(3))) const void*)(etab))));
#pragma unroll
                for (int p0 = 0; p0 < NC * E; p0 += 2) {
                    double t[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int p = p0 + q < NC * E ? p0 + q : p0;
                        t[q] = zn[p / E][p % E] + hx;
#pragma unroll
                        for (int f = 0; f < D; ++f) t[q] = fma(z[p / E][p % E][f], row.x[f], t[q]);
                    }
                    if (p0 + 1 < NC * E) {
                        tvs_exp2_pair<NTAB>(t[0], t[1], tab_addr, kv[p0], kv[p0 + 1]);
                        // the block has waited for every LDS and scalar load in flight, which the compiler cannot see inside it: say so (a wait
                        // that is already satisfied) -- otherwise its own bookkeeping waits for the row request at the head of the next step
                        __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0)
                    } else kv[p0] = tvs_exp2<NTAB>(t[0], etab);
                }
            }
        } else if constexpr (tvs_is_matern(KIND)) {
            constexpr int NTAB = tvs_etab_n(E == 2);
            constexpr int NRES = NTAB == 32 ? 1024 : NTAB;                                      // (the two-level A/B table has the 1,024-entry one's resolution)
            constexpr double S = tvs_matern_prescale(KIND, E == 2), K = TVS_LN2 / NRES;       // u = c r = q K
            const unsigned tab_addr = __builtin_amdgcn_readfirstlane(unsigned(uintptr_t((__attribute__((address_space(3))) const void*)(etab))));
#pragma unroll
            for (int p0 = 0; p0 < NC * E; p0 += 2) {
                double g[2], ev[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int p = p0 + q < NC * E ? p0 + q : p0;
                    double d = zn[p / E][p % E] + row.xs;                                      // _square_dist, kernels.py:765-776
#pragma unroll
                    for (int f = 0; f < D; ++f) d = fma(z[p / E][p % E][f], row.x[f], d);
                    d = fmax(d, 1e-40 * S * S);                                                // _euclid_dist, :779-781
                    const double y = __builtin_amdgcn_rsq(d), h = 0.5 * y;
                    double r = d * y;
                    r = fma(fma(-r, r, d), h, r);                                              // one Newton step on the residual: ~1.5 ulp
                    g[q] = r;
                }
                if constexpr (NTAB == 32 || NTAB == 64) {
                    ev[0] = tvs_exp2<NTAB>(-g[0], etab);
                    if (p0 + 1 < NC * E) ev[1] = tvs_exp2<NTAB>(-g[1], etab);
                } else if (p0 + 1 < NC * E) {
                    tvs_exp2_pair<NTAB, true>(g[0], g[1], tab_addr, ev[0], ev[1]);            // 2^(-q / N)
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                } else ev[0] = tvs_exp2<NTAB>(-g[0], etab);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (p0 + q >= NC * E) continue;
                    if constexpr (KIND == BASE_MATERN12) kv[p0 + q] = ev[q];                                        // :955-958
                    else if constexpr (KIND == BASE_MATERN32) kv[p0 + q] = fma(g[q], K, 1.0) * ev[q];               // :974-977
                    else kv[p0 + q] = fma(fma(g[q], K * K / 3.0, K), g[q], 1.0) * ev[q];                           // :991-993: 1 + u + u^2 / 3
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    double ip = z[c][e][0] * row.x[0];
#pragma unroll
                    for (int f = 1; f < D; ++f) ip = fma(z[c][e][f], row.x[f], ip);
                    kv[c * E + e] = ip;
                    a2[c * E + e] = zn[c][e];
                }
            if constexpr (KIND != BASE_LINEAR) tvs_base_eval_rest<NC * E>(A.kind, kv, a2, row.xs, A.p0, A.p1);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) k1[c] = E == 2 ? kv[c * E + 1] - kv[c * E] : kv[c * E];       // kernels.py:329-330
    }

    // one time step of the chains of this wave's levels (signature_algs.py:120-124)
    // HO (round 6): the higher-order chains of signature_algs.py:129-160 at the run-time order A.order >= 2 -- per component the vector r_j[l] of
    // repeat counts: r_j[0] = m_j U_{j-1} (U: the running totals, as at first order), r_j[l] = m_j r_{j-1}[l-1] / (l+1) of the SAME time step
    // (:153-155), U_j += sum_l r_j[l]; the state between steps is what first order keeps.
    __device__ __forceinline__ void chains(const double (&dm)[NC], int order) {
#pragma unroll
        for (int i = 1; i <= M; ++i) {
            if (!((MASK >> i) & 1)) continue;
            const int c0 = tvs_local_off(MASK, i);
            if constexpr (!HO) {
#pragma unroll
                for (int j = i - 1; j >= 1; --j) u[c0 + j] = fma(dm[c0 + j], u[c0 + j - 1], u[c0 + j]);
                u[c0] += dm[c0];
            } else {
                double rp[M], uold = u[c0];
                rp[0] = dm[c0];
                u[c0] += dm[c0];
#pragma unroll
                for (int j = 1; j < i; ++j) {
                    const double m = dm[c0 + j];
                    double rc[M], tot = m * uold;
                    rc[0] = tot;
#pragma unroll
                    for (int l = 1; l <= j; ++l) {
                        rc[l] = l < order ? (m * (1.0 / double(l + 1))) * rp[l - 1] : 0.0;        // :155 (d = min(j + 1, order))
                        tot += rc[l];
                    }
                    uold = u[c0 + j];
                    u[c0 + j] += tot;
#pragma unroll
                    for (int l = 0; l <= j; ++l) rp[l] = rc[l];
                }
            }
        }
    }

    // One sequence: rows g0 .. g0 + L - 1 of the row array.  `cur` holds row g0 on entry and row g0 + L -- the first row of the NEXT sequence, the
    // array is contiguous and one row longer than the sequences -- on exit: every step asks for the row after its own before it evaluates, so the
    // scalar loads have a step's arithmetic to arrive in.
    __device__ __forceinline__ void sweep(const TvsTileArgs& A, tvs_cptr rows, int64_t g0, const double* __restrict__ etab, Row& cur) {
        const int L = A.L;
        double ka[NC], kb[NC], dm[NC];
        Row nxt;
        if (!A.difference) {
            for (int tau = 0; tau < L; ++tau) {
                nxt = tvs_load_row<D>(rows, g0 + tau + 1);
                __builtin_amdgcn_sched_barrier(0);            // (the request stays at the head of the step)
                eval(A, cur, etab, ka);
                chains(ka, A.order);
                cur = nxt;
            }
        } else if constexpr (KIND == BASE_LINEAR) {               // rows are increments already (row 0 unused)
            cur = tvs_load_row<D>(rows, g0 + 1);                  // (L == 1: the next sequence's row 0, as promised)
            for (int tau = 1; tau < L; ++tau) {
                nxt = tvs_load_row<D>(rows, g0 + tau + 1);
                __builtin_amdgcn_sched_barrier(0);            // (the request stays at the head of the step)
                eval(A, cur, etab, ka);
                chains(ka, A.order);
                cur = nxt;
            }
        } else {                                                  // signature_algs.py:114: difference along time
            nxt = tvs_load_row<D>(rows, g0 + 1);
            eval(A, cur, etab, ka);
            cur = nxt;
            int tau = 1;
            for (; tau + 1 < L; tau += 2) {                       // two steps per trip: the previous values alternate registers
                nxt = tvs_load_row<D>(rows, g0 + tau + 1);
                __builtin_amdgcn_sched_barrier(0);            // (the request stays at the head of the step)
                eval(A, cur, etab, kb);
#pragma unroll
                for (int c = 0; c < NC; ++c) dm[c] = kb[c] - ka[c];
                chains(dm, A.order);
                cur = nxt;
                nxt = tvs_load_row<D>(rows, g0 + tau + 2);
                __builtin_amdgcn_sched_barrier(0);
                eval(A, cur, etab, ka);
#pragma unroll
                for (int c = 0; c < NC; ++c) dm[c] = ka[c] - kb[c];
                chains(dm, A.order);
                cur = nxt;
            }
            if (tau < L) {
                nxt = tvs_load_row<D>(rows, g0 + tau + 1);
                __builtin_amdgcn_sched_barrier(0);            // (the request stays at the head of the step)
                eval(A, cur, etab, kb);
#pragma unroll
                for (int c = 0; c < NC; ++c) dm[c] = kb[c] - ka[c];
                chains(dm, A.order);
                cur = nxt;
            }
        }
    }

    // the chain totals of the sequence just swept, for the reverse pass: consecutive lanes (tensors) write consecutive addresses
    __device__ __forceinline__ void store_aux(const TvsTileArgs& A, int64_t n, int64_t t) const {
        constexpr int lt = M * (M + 1) / 2;
#pragma unroll
        for (int i = 1; i <= M; ++i) {
            if (!((MASK >> i) & 1)) continue;
#pragma unroll
            for (int j = 0; j < i; ++j) A.aux[(n * lt + i * (i - 1) / 2 + j) * A.Tpad + t] = u[tvs_local_off(MASK, i) + j];
        }
    }

    // weighted levels of the sequence just swept into column `col` of the wave's tile; resets the chains.  Level sum: the first set writes
    // (level 0 with it), the others add -- a fixed order.  Level arrays: slot s of the tile holds the s-th level of this set (level 0 first).
    __device__ __forceinline__ void emit(const TvsTileArgs& A, const double (&fac)[M + 1], double* __restrict__ tile, int lane, int col, bool first) {
        constexpr int TS = TVS_TILE_S + 1;
        if (A.sum_levels) {
            double acc = first ? fac[0] : tile[lane * TS + col];      // level 0 == 1 (signature_algs.py:116)
#pragma unroll
            for (int i = 1; i <= M; ++i)
                if ((MASK >> i) & 1) acc += u[tvs_local_off(MASK, i) + i - 1] * fac[i];
            tile[lane * TS + col] = acc;
        } else {
            int slot = 0;
            if (first) tile[(slot++ * 64 + lane) * TS + col] = fac[0];
#pragma unroll
            for (int i = 1; i <= M; ++i)
                if ((MASK >> i) & 1) tile[(slot++ * 64 + lane) * TS + col] = u[tvs_local_off(MASK, i) + i - 1] * fac[i];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) u[c] = 0.0;
    }
};

// A lane's state in doubles: per component E points of D features + a squared norm each, the chain value and a previous kernel value (the
// sequence's row is wave-uniform: scalar registers).  The kernel is compiled for two wavefronts per SIMD (256 registers): three (the 168
// registers a state of <= 70 doubles fits: M = 4 with increments in three sets) measured the SAME time as two with the levels in two sets -- 5.74
// against 5.74 ms at configs[2] -- and fewer sets win where they fit (one set 2.66, two 2.98, three 3.09 ms without increments:
// profiles/r05_bench_c3_variants.txt): every set sweeps the sequence again, and a float64-dense kernel runs at the chip's power limit from two
// wavefronts per SIMD on (tools/microbench4.hip).
constexpr int tvs_state_doubles(int M, int P, int D, bool incr, int kind) {
    const int E = (incr && kind != BASE_LINEAR) ? 2 : 1;
    return tvs_max_comps(M, P) * (E * (D + 1) + (kind == BASE_LINEAR ? 0 : 3));
}
constexpr int tvs_waves_per_simd(int, int, int, bool, int) { return 2; }
// which numbers of level sets are built for num_levels = M (tvs_tile_inst_m*.hip)
constexpr bool tvs_built_sets(int M, int P) { return M == 2 ? P == 1 : ((M == 3 || M == 4) ? (P == 1 || P == 2) : ((M == 5 || M == 6) && (P == 2 || P == 3))); }
// Level sets the planner takes: the fewest whose largest set compiles without spills inside the sweep at two wavefronts per SIMD.  Limits read off
// the compiler's register reports: 100 doubles of lane state for the families fixed at compile time (RBF, M = 4, D = 6: one set of 10 components =
// 100 doubles = 227 registers, no scratch; with increments two sets of 5 = 85 doubles), 90 for the families evaluated through base_eval_n at run
// time (they spill some tens of registers there and are still faster than the older kernels).  0: no built variant fits.
constexpr int tvs_planned_sets(int M, int D, bool incr, int kind) {
    const int limit = kind == -1 ? 90 : 100;
    for (int P = 1; P <= 3; ++P) {
        if (P > 1 && M < 3) break;
        if (tvs_built_sets(M, P) && tvs_state_doubles(M, P, D, incr, kind) <= limit) return P;
    }
    return 0;
}

template <int M, int P, int D, bool INCR, int KIND, bool HO = false>
__global__ __launch_bounds__(TVS_WG_WAVES * 64, tvs_waves_per_simd(M, P, D, INCR, KIND)) void tvs_tile_kernel(const TvsTileArgs A) {
    constexpr int TS = TVS_TILE_S + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char tvs_tile_smem[];
    double* const etab = reinterpret_cast<double*>(tvs_tile_smem);
    constexpr int NTAB = tvs_etab_n(INCR && KIND != BASE_LINEAR);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nslots = A.sum_levels ? 1 : tvs_tile_slots(M, P, false);
    double* const tile = etab + tvs_etab_doubles(INCR && KIND != BASE_LINEAR) + size_t(wave) * nslots * 64 * TS;      // this wave's [slots][64][TS]
    const int TB = int(A.Tpad / 64);
    tvs_cptr rows = (tvs_cptr)(A.XR);
    tvs_cptr fx = (tvs_cptr)(A.fx);
    tvs_cptr wts = (tvs_cptr)(A.w);
    double* __restrict__ out = static_cast<double*>(A.out);

    if constexpr (KIND != BASE_LINEAR) {
        if constexpr (NTAB == 32) exp_tab2l_fill(etab, tid, TVS_WG_WAVES * 64);
        else if constexpr (NTAB == 64) exp_tab_fill(etab, tid, TVS_WG_WAVES * 64);
        else if constexpr (NTAB == 256) exp_tab256_fill(etab, tid, TVS_WG_WAVES * 64);
        else exp_tabn_fill<NTAB>(etab, tid, TVS_WG_WAVES * 64);
        __syncthreads();                                          // the only barrier: from here on the waves go their own ways
    }

    // this wave's next item of tensor block tb: one lane asks; the answer is read (and waited for) where it is needed
    auto ask = [&](int tb) {
        int v = 0;
        if (lane == 0) v = atomicAdd(A.queue + tb, 1);
        return v;
    };
    // levels of the tile's columns 0 .. ncols-1, tile slot `slot` -> level array lv of the result: 4 tensor rows x 16 sequences per store
    auto flush = [&](int slot, int lv, int tb, int64_t nb, int ncols) {
        __builtin_amdgcn_wave_barrier();
        const int r4 = lane >> 4, cc = lane & 15;
#pragma unroll 4
        for (int r = r4; r < 64; r += 4) {
            const int64_t tr = tb * int64_t(64) + r;
            if (tr < A.Tn && cc < ncols) out[(int64_t(lv) * A.Tn + tr) * A.N + nb + cc] = tile[(slot * 64 + r) * TS + cc];
        }
        __builtin_amdgcn_wave_barrier();
    };
    // one set of levels over the sequences n0 .. n1-1 (a tile's columns)
    auto sweep_set = [&](auto& W, bool first, int tb, int64_t t, int64_t n0, int64_t n1) {
        TvsRow<D> row = tvs_load_row<D>(rows, n0 * A.L);
        for (int64_t n = n0; n < n1; ++n) {
            W.sweep(A, rows, n * A.L, etab, row);
            if (A.aux) W.store_aux(A, n, t);
            double fac[M + 1];
#pragma unroll
            for (int i = 0; i <= M; ++i) {
                double f = fx ? fx[n * (M + 1) + i] : 1.0;
                if (wts) f *= wts[i];
                fac[i] = f;
            }
            W.emit(A, fac, tile, lane, int(n - n0), first);
        }
    };
    auto flush_levels = [&](int mask, bool first, int tb, int64_t n0, int ncols) {
        int slot = 0;
        if (first) flush(slot++, 0, tb, n0, ncols);
        for (int i = 1; i <= M; ++i)
            if ((mask >> i) & 1) flush(slot++, i, tb, n0, ncols);
    };

    // a wave starts at tensor block (its index mod TB) and moves on to the next block when that one's queue is empty
    const unsigned worker = blockIdx.x * TVS_WG_WAVES + wave;
    for (int q = 0; q < TB; ++q) {
        const int tb = (int(worker % unsigned(TB)) + q) % TB;
        int it = __builtin_amdgcn_readfirstlane(ask(tb));
        if (it >= A.items) continue;
        const int64_t t = tb * int64_t(64) + lane;                // < Tpad
        TvsTileWave<M, P, D, INCR, KIND, tvs_level_mask(M, P, 0), HO> W0;
        if constexpr (P == 1) W0.load(A, t);                      // one set: the components stay in registers across the items
        while (it < A.items) {
            int64_t n_begin, n_end;
            A.item(it, &n_begin, &n_end);
            const int nx = ask(tb);                               // the item after this one, asked for now
            for (int64_t n0 = n_begin; n0 < n_end; n0 += TVS_TILE_S) {
                const int64_t n1 = n0 + TVS_TILE_S < n_end ? n0 + TVS_TILE_S : n_end;
                if constexpr (P > 1) W0.load(A, t);
                sweep_set(W0, true, tb, t, n0, n1);
                if (!A.sum_levels) flush_levels(tvs_level_mask(M, P, 0), true, tb, n0, int(n1 - n0));
                if constexpr (P > 1) {
                    TvsTileWave<M, P, D, INCR, KIND, tvs_level_mask(M, P, 1), HO> W1;
                    W1.load(A, t);
                    sweep_set(W1, false, tb, t, n0, n1);
                    if (!A.sum_levels) flush_levels(tvs_level_mask(M, P, 1), false, tb, n0, int(n1 - n0));
                }
                if constexpr (P > 2) {
                    TvsTileWave<M, P, D, INCR, KIND, tvs_level_mask(M, P, P > 2 ? 2 : 0), HO> W2;
                    W2.load(A, t);
                    sweep_set(W2, false, tb, t, n0, n1);
                    if (!A.sum_levels) flush_levels(tvs_level_mask(M, P, P > 2 ? 2 : 0), false, tb, n0, int(n1 - n0));
                }
                if (A.sum_levels) flush(0, 0, tb, n0, int(n1 - n0));
            }
            it = __builtin_amdgcn_readfirstlane(nx);
        }
    }
}

// Prepared inducing tensors in the tensor-lane layout.  In: Z (lt, T, E_in, d_eff).  Out:
//   ZL[((k * E + e) * D + fe) * Tpad + t] = pre * z~ (zero for fe >= d_eff),   ZN[(k * E + e) * Tpad + t] = |pre * z~|^2   (zero for t >= T)
// with z~ the scaled component (kernels.py:367-398) and, for collapse (linear kernel, E_in = 2, E = 1), the difference of the
// component's two points (kernels.py:329-330 applied before the inner product, which is linear in it).
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void prep_tensors_tile_kernel(const double* __restrict__ Z, int lt, int64_t Tn, int64_t Tpad, int E_in, int collapse,
                                         double pre, ScaleParams P, int D, double* __restrict__ ZL, double* __restrict__ ZN,
                                         int32_t* __restrict__ queue) {
    const int d_eff = P.d_eff();
    if (blockIdx.x == 0 && queue)                                 // the item counters of the tile kernel's launch behind this one
        for (int64_t b = threadIdx.x; b < Tpad / 64; b += blockDim.x) queue[b] = 0;
    const int E = collapse ? 1 : E_in;
    const int64_t total = Tpad * lt * E;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t t = idx % Tpad;
        const int e = int((idx / Tpad) % E);
        const int k = int(idx / (Tpad * E));
        double ss = 0.0;
        for (int fe = 0; fe < d_eff; ++fe) {
            const int lag = fe / P.d_in, f = fe - lag * P.d_in;
            double v = 0.0;
            if (t < Tn) {
                const double* zp = Z + ((int64_t(k) * Tn + t) * E_in + e) * d_eff + fe;
                v = collapse ? zp[d_eff] - zp[0] : zp[0];
                if (P.has_ls) {
                    v = v / P.lsv(f);
                    if (P.num_lags > 0) v = v * P.gamma[lag];
                }
                v *= pre;
            }
            ZL[((int64_t(k) * E + e) * D + fe) * Tpad + t] = v;
            ss = fma(v, v, ss);
        }
        for (int fe = d_eff; fe < D; ++fe) ZL[((int64_t(k) * E + e) * D + fe) * Tpad + t] = 0.0;
        ZN[(int64_t(k) * E + e) * Tpad + t] = ss;
    }
}
#else
__global__ void prep_tensors_tile_kernel(const double* __restrict__ Z, int lt, int64_t Tn, int64_t Tpad, int E_in, int collapse,
                                         double pre, ScaleParams P, int D, double* __restrict__ ZL, double* __restrict__ ZN,
                                         int32_t* __restrict__ queue);
#endif

// Records of the sequences: rec[n][tau * D + fe] = pre * x~[n][tau][fe]  (increments == 1: x~[tau] - x~[tau-1], row 0
// zero; columns fe >= d_eff zero), then rec[n][L * D + tau] = |pre * x~[n][tau]|^2; the tail up to rec_elems is zero-filled here.
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void prep_seq_tile_records_kernel(const double* __restrict__ X, int64_t N, int L, ScaleParams P, double pre,
                                             int increments, int D, int rec_elems, double* __restrict__ out) {
    const int d_eff = P.d_eff();
    const int64_t total = N * int64_t(rec_elems);
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t n = idx / rec_elems;
        const int q = int(idx - n * rec_elems);
        const double* Xn = X + n * int64_t(L) * P.d_in;
        double v = 0.0;
        if (q < L * D) {
            const int tau = q / D, fe = q - tau * D;
            if (fe >= d_eff) v = 0.0;
            else if (!increments) v = pre * scaled_point<double>(Xn, L, tau, fe, P);
            else if (tau >= 1) v = pre * (scaled_point<double>(Xn, L, tau, fe, P) - scaled_point<double>(Xn, L, tau - 1, fe, P));
        } else if (q < L * D + L) {
            const int tau = q - L * D;
            for (int fe = 0; fe < d_eff; ++fe) {
                const double s = pre * scaled_point<double>(Xn, L, tau, fe, P);
                v = fma(s, s, v);
            }
        }
        out[idx] = v;
    }
}
#else
__global__ void prep_seq_tile_records_kernel(const double* __restrict__ X, int64_t N, int L, ScaleParams P, double pre,
                                             int increments, int D, int rec_elems, double* __restrict__ out);
#endif

// Rows of the sequences for the forward tile kernel (TvsTileArgs::XR): row[n * L + tau][fe] = pre * x~[n][tau][fe]  (increments == 1:
// x~[tau] - x~[tau-1], row 0 of a sequence zero; columns fe >= d_eff zero), row[.][D] = |pre * x~[n][tau]|^2, the rest of the RS doubles and the one
// extra row behind the last sequence zero.
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void prep_seq_tile_rows_kernel(const double* __restrict__ X, int64_t N, int L, ScaleParams P, double pre,
                                                 int increments, int D, int RS, double* __restrict__ out) {
    const int d_eff = P.d_eff();
    const int64_t total = (N * int64_t(L) + 1) * RS;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t g = idx / RS;
        const int q = int(idx - g * RS);
        double v = 0.0;
        if (g < N * int64_t(L)) {
            const int64_t n = g / L;
            const int tau = int(g - n * L);
            const double* Xn = X + n * int64_t(L) * P.d_in;
            if (q < d_eff) {
                if (!increments) v = pre * scaled_point<double>(Xn, L, tau, q, P);
                else if (tau >= 1) v = pre * (scaled_point<double>(Xn, L, tau, q, P) - scaled_point<double>(Xn, L, tau - 1, q, P));
            } else if (q == D) {
                for (int fe = 0; fe < d_eff; ++fe) {
                    const double s = pre * scaled_point<double>(Xn, L, tau, fe, P);
                    v = fma(s, s, v);
                }
            }
        }
        out[idx] = v;
    }
}
#else
__global__ void prep_seq_tile_rows_kernel(const double* __restrict__ X, int64_t N, int L, ScaleParams P, double pre,
                                                 int increments, int D, int RS, double* __restrict__ out);
#endif

}  // namespace gpsig
