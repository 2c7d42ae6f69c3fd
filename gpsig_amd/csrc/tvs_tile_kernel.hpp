// tvs_tile_kernel.hpp -- inducing tensors vs sequences (Kzx) for many tensors: SignatureKernel._K_tens_vs_seq +
// signature_kern_tens_vs_seq_first_order (gpsig/kernels.py:313-340, gpsig/signature_algs.py:101-127) + the epilogue of
// K_tens_vs_seq (kernels.py:572-588), order 1.
//
// Mapping.  A workgroup owns a block of 64 inducing tensors (lane = tensor) and a RUN of consecutive sequences.  Its NW
// wavefronts split the LEVELS between them (a level's chain only involves its own components, signature_algs.py:118-125):
// wave w keeps the components of its levels in registers for the whole run -- M = 4 with increments is 2 waves x 5
// components x 2 points x d doubles, which fits 2-3 waves per SIMD where all 10 components in one lane would not fit one.
// The sequence is staged ONCE per workgroup into LDS by LDS-DMA (global_load_lds, double-buffered: sequence n+1 lands
// while n is swept) and read back as same-address broadcasts; one sweep over time per sequence with the running sums of
// the chains in registers,
//     u[k0+j] += dM[k0+j](tau) * u[k0+j-1]   (j = i-1 .. 1, old values),   u[k0] += dM[k0](tau),   K_i = u[k0+i-1],
// no cross-lane traffic.  Results are collected in an LDS tile [tensor][sequence] and written out 16 sequences at a time,
// so that every store instruction writes four full 128-byte lines of the (T, N) result (the previous tensor-lane kernel
// stored one 8-byte value per lane at a stride of N).
//
// Base kernel at compile time: KIND = BASE_LINEAR (records hold increments of the scaled sequence when difference is on,
// tensors with increments are collapsed to z1 - z0 on the way in: one inner product per component and time step),
// KIND = BASE_RBF (points prepared in units of sqrt(ln2/256), so the inner products are the argument of the table-driven
// 2^(t/256), fast_exp.hpp), or KIND = -1: the family is a run-time value (base_eval_n of seq_core.hpp).
#pragma once

#include "aux_kernels.hpp"
#include "fast_exp.hpp"
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
    // the item schedule for `wgs` workgroups sharing one tensor block: 70 % of the sequences in runs of r, 20 % in runs of r/2, the rest in runs of r/4
    void plan_items(int64_t wgs) {
        const int64_t share = (N + wgs - 1) / (wgs < 1 ? 1 : wgs);
        int r = share >= 96 ? 32 : (share >= 48 ? 16 : (share >= 24 ? 8 : 4));
        run = r; run2 = r / 2 > 0 ? r / 2 : 1; run3 = r / 4 > 0 ? r / 4 : 1;
        cnt1 = int32_t((N * 7 / 10) / run);
        cnt2 = int32_t((N * 2 / 10) / run2);
        const int64_t rest = N - int64_t(cnt1) * run - int64_t(cnt2) * run2;
        items = cnt1 + cnt2 + int32_t((rest + run3 - 1) / run3);
    }
};

// LDS bytes of one workgroup: the exp table, the result tile and the item slot
inline size_t tvs_tile_lds_bytes(int M, int NW, bool sum_levels, bool two_points) {
    const size_t slots = sum_levels ? size_t(NW) : size_t(M + 1);
    return sizeof(double) * (tvs_etab_doubles(two_points) + slots * 64 * (TVS_TILE_S + 1) + 2);
}

// ---- The table-driven 2^(t/N) of fast_exp.hpp (kexp2_tabn / kexp2_tab256: same operations, same order, same bits) for TWO arguments at once, as one
// block of hand-scheduled instructions: both roundings and both table reads first, the polynomial tails while the reads are in flight, one wait.
// Left to the compiler each exp is one dependent chain that issues its read behind its tail and waits for it at once (and any attempt to steer it
// with sched_barrier spilled the tensors' components).  11 vector instructions per exp.  tab: LDS byte address of the table (a scalar).
template <int N>
__device__ __forceinline__ void tvs_exp2_pair(double t0, double t1, unsigned tab, double& e0, double& e1) {
    static_assert(N == 256 || N == 1024 || N == 2048, "table sizes with an asm form");
    double r0, r1, q0, q1;
    int i0, i1, a0, a1;
    if constexpr (N == 256) {
        const double c4 = 0x1.3b2ab6fba4e77p-39, c3 = 0x1.c6b08d704a0c0p-29, c2 = 0x1.ebfbdff82c58fp-19, c1 = 0x1.62e42fefa39efp-9;
        asm volatile(
            "v_rndne_f64 %[r0], %[t0]\n\tv_rndne_f64 %[r1], %[t1]\n\t"
            "v_cvt_i32_f64 %[i0], %[r0]\n\tv_cvt_i32_f64 %[i1], %[r1]\n\t"
            "v_bfe_u32 %[a0], %[i0], 0, 8\n\tv_bfe_u32 %[a1], %[i1], 0, 8\n\t"
            "v_lshl_add_u32 %[a0], %[a0], 3, %[tab]\n\tv_lshl_add_u32 %[a1], %[a1], 3, %[tab]\n\t"
            "ds_read_b64 %[e0], %[a0]\n\tds_read_b64 %[e1], %[a1]\n\t"
            "v_add_f64 %[r0], %[t0], -%[r0]\n\tv_add_f64 %[r1], %[t1], -%[r1]\n\t"
            "v_fma_f64 %[q0], %[c4], %[r0], %[c3]\n\tv_fma_f64 %[q1], %[c4], %[r1], %[c3]\n\t"
            "v_fma_f64 %[q0], %[q0], %[r0], %[c2]\n\tv_fma_f64 %[q1], %[q1], %[r1], %[c2]\n\t"
            "v_fma_f64 %[q0], %[q0], %[r0], %[c1]\n\tv_fma_f64 %[q1], %[q1], %[r1], %[c1]\n\t"
            "v_mul_f64 %[r0], %[q0], %[r0]\n\tv_mul_f64 %[r1], %[q1], %[r1]\n\t"
            "v_ashrrev_i32 %[i0], 8, %[i0]\n\tv_ashrrev_i32 %[i1], 8, %[i1]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_fma_f64 %[e0], %[e0], %[r0], %[e0]\n\tv_fma_f64 %[e1], %[e1], %[r1], %[e1]\n\t"
            "v_ldexp_f64 %[e0], %[e0], %[i0]\n\tv_ldexp_f64 %[e1], %[e1], %[i1]"
            : [e0] "=&v"(e0), [e1] "=&v"(e1), [r0] "=&v"(r0), [r1] "=&v"(r1), [q0] "=&v"(q0), [q1] "=&v"(q1), [i0] "=&v"(i0), [i1] "=&v"(i1),
              [a0] "=&v"(a0), [a1] "=&v"(a1)
            : [t0] "v"(t0), [t1] "v"(t1), [tab] "s"(tab), [c4] "s"(c4), [c3] "v"(c3), [c2] "s"(c2), [c1] "s"(c1));
    } else {
        constexpr double c3 = ExpTabN<N>::C3, c2 = ExpTabN<N>::C2, c1 = ExpTabN<N>::C1;
        const double c3s = c3, c2v = c2, c1s = c1;
        if constexpr (N == 1024)
            asm volatile(
                "v_rndne_f64 %[r0], %[t0]\n\tv_rndne_f64 %[r1], %[t1]\n\t"
                "v_cvt_i32_f64 %[i0], %[r0]\n\tv_cvt_i32_f64 %[i1], %[r1]\n\t"
                "v_bfe_u32 %[a0], %[i0], 0, 10\n\tv_bfe_u32 %[a1], %[i1], 0, 10\n\t"
                "v_lshl_add_u32 %[a0], %[a0], 3, %[tab]\n\tv_lshl_add_u32 %[a1], %[a1], 3, %[tab]\n\t"
                "ds_read_b64 %[e0], %[a0]\n\tds_read_b64 %[e1], %[a1]\n\t"
                "v_add_f64 %[r0], %[t0], -%[r0]\n\tv_add_f64 %[r1], %[t1], -%[r1]\n\t"
                "v_fma_f64 %[q0], %[c3], %[r0], %[c2]\n\tv_fma_f64 %[q1], %[c3], %[r1], %[c2]\n\t"
                "v_fma_f64 %[q0], %[q0], %[r0], %[c1]\n\tv_fma_f64 %[q1], %[q1], %[r1], %[c1]\n\t"
                "v_mul_f64 %[r0], %[q0], %[r0]\n\tv_mul_f64 %[r1], %[q1], %[r1]\n\t"
                "v_ashrrev_i32 %[i0], 10, %[i0]\n\tv_ashrrev_i32 %[i1], 10, %[i1]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_fma_f64 %[e0], %[e0], %[r0], %[e0]\n\tv_fma_f64 %[e1], %[e1], %[r1], %[e1]\n\t"
                "v_ldexp_f64 %[e0], %[e0], %[i0]\n\tv_ldexp_f64 %[e1], %[e1], %[i1]"
                : [e0] "=&v"(e0), [e1] "=&v"(e1), [r0] "=&v"(r0), [r1] "=&v"(r1), [q0] "=&v"(q0), [q1] "=&v"(q1), [i0] "=&v"(i0), [i1] "=&v"(i1),
                  [a0] "=&v"(a0), [a1] "=&v"(a1)
                : [t0] "v"(t0), [t1] "v"(t1), [tab] "s"(tab), [c3] "s"(c3s), [c2] "v"(c2v), [c1] "s"(c1s));
        else
            asm volatile(
                "v_rndne_f64 %[r0], %[t0]\n\tv_rndne_f64 %[r1], %[t1]\n\t"
                "v_cvt_i32_f64 %[i0], %[r0]\n\tv_cvt_i32_f64 %[i1], %[r1]\n\t"
                "v_bfe_u32 %[a0], %[i0], 0, 11\n\tv_bfe_u32 %[a1], %[i1], 0, 11\n\t"
                "v_lshl_add_u32 %[a0], %[a0], 3, %[tab]\n\tv_lshl_add_u32 %[a1], %[a1], 3, %[tab]\n\t"
                "ds_read_b64 %[e0], %[a0]\n\tds_read_b64 %[e1], %[a1]\n\t"
                "v_add_f64 %[r0], %[t0], -%[r0]\n\tv_add_f64 %[r1], %[t1], -%[r1]\n\t"
                "v_fma_f64 %[q0], %[c3], %[r0], %[c2]\n\tv_fma_f64 %[q1], %[c3], %[r1], %[c2]\n\t"
                "v_fma_f64 %[q0], %[q0], %[r0], %[c1]\n\tv_fma_f64 %[q1], %[q1], %[r1], %[c1]\n\t"
                "v_mul_f64 %[r0], %[q0], %[r0]\n\tv_mul_f64 %[r1], %[q1], %[r1]\n\t"
                "v_ashrrev_i32 %[i0], 11, %[i0]\n\tv_ashrrev_i32 %[i1], 11, %[i1]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_fma_f64 %[e0], %[e0], %[r0], %[e0]\n\tv_fma_f64 %[e1], %[e1], %[r1], %[e1]\n\t"
                "v_ldexp_f64 %[e0], %[e0], %[i0]\n\tv_ldexp_f64 %[e1], %[e1], %[i1]"
                : [e0] "=&v"(e0), [e1] "=&v"(e1), [r0] "=&v"(r0), [r1] "=&v"(r1), [q0] "=&v"(q0), [q1] "=&v"(q1), [i0] "=&v"(i0), [i1] "=&v"(i1),
                  [a0] "=&v"(a0), [a1] "=&v"(a1)
                : [t0] "v"(t0), [t1] "v"(t1), [tab] "s"(tab), [c3] "s"(c3s), [c2] "v"(c2v), [c1] "s"(c1s));
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

template <int M, int NW, int D, bool INCR, int KIND, int MASK>
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
                    zn[c][e] = KIND == BASE_RBF ? -0.5 * s : s;
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
            if constexpr (KIND != BASE_LINEAR) base_eval_n<double, NC * E>(A.kind, kv, a2, row.xs, A.p0, A.p1);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) k1[c] = E == 2 ? kv[c * E + 1] - kv[c * E] : kv[c * E];       // kernels.py:329-330
    }

    // one time step of the chains of this wave's levels (signature_algs.py:120-124)
    __device__ __forceinline__ void chains(const double (&dm)[NC]) {
#pragma unroll
        for (int i = 1; i <= M; ++i) {
            if (!((MASK >> i) & 1)) continue;
            const int c0 = tvs_local_off(MASK, i);
#pragma unroll
            for (int j = i - 1; j >= 1; --j) u[c0 + j] = fma(dm[c0 + j], u[c0 + j - 1], u[c0 + j]);
            u[c0] += dm[c0];
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
                chains(ka);
                cur = nxt;
            }
        } else if constexpr (KIND == BASE_LINEAR) {               // rows are increments already (row 0 unused)
            cur = tvs_load_row<D>(rows, g0 + 1);                  // (L == 1: the next sequence's row 0, as promised)
            for (int tau = 1; tau < L; ++tau) {
                nxt = tvs_load_row<D>(rows, g0 + tau + 1);
                __builtin_amdgcn_sched_barrier(0);            // (the request stays at the head of the step)
                eval(A, cur, etab, ka);
                chains(ka);
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
                chains(dm);
                cur = nxt;
                nxt = tvs_load_row<D>(rows, g0 + tau + 2);
                __builtin_amdgcn_sched_barrier(0);
                eval(A, cur, etab, ka);
#pragma unroll
                for (int c = 0; c < NC; ++c) dm[c] = ka[c] - kb[c];
                chains(dm);
                cur = nxt;
            }
            if (tau < L) {
                nxt = tvs_load_row<D>(rows, g0 + tau + 1);
                __builtin_amdgcn_sched_barrier(0);            // (the request stays at the head of the step)
                eval(A, cur, etab, kb);
#pragma unroll
                for (int c = 0; c < NC; ++c) dm[c] = kb[c] - ka[c];
                chains(dm);
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

    // weighted levels of the sequence just swept into the tile column `col`; resets the chains
    __device__ __forceinline__ void emit(const TvsTileArgs& A, const double (&fac)[M + 1], double* __restrict__ tile, int wave,
                                         int lane, int col, bool with_level0) {
        constexpr int TS = TVS_TILE_S + 1;
        double acc = 0.0;
        if (with_level0) {                                        // level 0 == 1 (signature_algs.py:116)
            if (A.sum_levels) acc = fac[0];
            else tile[(0 * 64 + lane) * TS + col] = fac[0];
        }
#pragma unroll
        for (int i = 1; i <= M; ++i) {
            if (!((MASK >> i) & 1)) continue;
            const double v = u[tvs_local_off(MASK, i) + i - 1] * fac[i];
            if (A.sum_levels) acc += v;
            else tile[(i * 64 + lane) * TS + col] = v;
        }
        if (A.sum_levels) tile[(wave * 64 + lane) * TS + col] = acc;
#pragma unroll
        for (int c = 0; c < NC; ++c) u[c] = 0.0;
    }
};

// Wavefronts per SIMD the kernel is compiled for.  A gfx950 SIMD issues a float64 instruction every 4 cycles only with THREE wavefronts to pick
// from -- two get one every 5.3, one every 8, however independent the instructions are (tools/clockcheck.hip, profiles/r03_clockcheck.txt) -- so a
// lane state that fits 168 registers is worth a third of the kernel's time.  A lane's state in doubles: per component E points of D features + a
// squared norm each, the chain value and a previous kernel value (the sequence's row is wave-uniform: scalar registers).
constexpr int tvs_state_doubles(int M, int NW, int D, bool incr, int kind) {
    const int E = (incr && kind != BASE_LINEAR) ? 2 : 1;
    return tvs_max_comps(M, NW) * (E * (D + 1) + (kind == BASE_LINEAR ? 0 : 3));
}
constexpr int tvs_waves_per_simd(int M, int NW, int D, bool incr, int kind) {
    return (kind != BASE_LINEAR && tvs_state_doubles(M, NW, D, incr, kind) <= 70) ? 3 : 2;
}

template <int M, int NW, int D, bool INCR, int KIND>
__global__ __launch_bounds__(NW * 64, tvs_waves_per_simd(M, NW, D, INCR, KIND)) void tvs_tile_kernel(const TvsTileArgs A) {
    constexpr int TS = TVS_TILE_S + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char tvs_tile_smem[];
    double* const etab = reinterpret_cast<double*>(tvs_tile_smem);
    constexpr int NTAB = tvs_etab_n(INCR && KIND != BASE_LINEAR);
    double* const tile = etab + tvs_etab_doubles(INCR && KIND != BASE_LINEAR);     // [slots][64][TS]
    const int nslots = A.sum_levels ? NW : M + 1;
    int* const sh_item = reinterpret_cast<int*>(tile + size_t(nslots) * 64 * TS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TB = int(A.Tpad / 64);
    tvs_cptr rows = (tvs_cptr)(A.XR);
    tvs_cptr fx = (tvs_cptr)(A.fx);
    tvs_cptr wts = (tvs_cptr)(A.w);
    double* __restrict__ out = static_cast<double*>(A.out);

    if constexpr (KIND != BASE_LINEAR) {
        if constexpr (NTAB == 32) exp_tab2l_fill(etab, tid, NW * 64);
        else if constexpr (NTAB == 64) exp_tab_fill(etab, tid, NW * 64);
        else if constexpr (NTAB == 256) exp_tab256_fill(etab, tid, NW * 64);
        else exp_tabn_fill<NTAB>(etab, tid, NW * 64);
    }

    // the workgroup's next item of tensor block tb: one lane asks, everybody reads the answer behind a barrier
    auto draw = [&](int tb) {
        if (tid == 0) sh_item[0] = atomicAdd(A.queue + tb, 1);
        __syncthreads();
        const int it = __builtin_amdgcn_readfirstlane(sh_item[0]);
        __syncthreads();
        return it;
    };

    auto run_wave = [&](auto& W) {
        // a workgroup starts at tensor block (its index mod TB) and moves on to the next block when that one's queue is empty
        for (int q = 0; q < TB; ++q) {
            const int tb = (int(blockIdx.x % unsigned(TB)) + q) % TB;
            int it = draw(tb);
            if (it >= A.items) continue;
            const int64_t t = tb * int64_t(64) + lane;            // < Tpad
            W.load(A, t);
            while (it < A.items) {
                int64_t n_begin, n_end;
                A.item(it, &n_begin, &n_end);
                int nxt = 0;
                if (tid == 0) nxt = atomicAdd(A.queue + tb, 1);   // the item after this one: the answer is not needed before this item's last flush
                TvsRow<D> row = tvs_load_row<D>(rows, n_begin * A.L);
                for (int64_t n = n_begin; n < n_end; ++n) {
                    const int idx = int(n - n_begin), col = idx % TVS_TILE_S;
                    W.sweep(A, rows, n * A.L, etab, row);
                    if (A.aux) W.store_aux(A, n, t);
                    double fac[M + 1];
#pragma unroll
                    for (int i = 0; i <= M; ++i) {
                        double f = fx ? fx[n * (M + 1) + i] : 1.0;
                        if (wts) f *= wts[i];
                        fac[i] = f;
                    }
                    W.emit(A, fac, tile, wave, lane, col, wave == 0);
                    if (col == TVS_TILE_S - 1 || n + 1 == n_end) {
                        if (n + 1 == n_end && tid == 0) sh_item[0] = nxt;
                        __syncthreads();                                  // every wave's columns are in the tile
                        // flush: 4 tensor rows x 16 sequences per store instruction
                        const int64_t nb = n - col;                       // first sequence of the tile
                        const int r4 = lane >> 4, cc = lane & 15;
                        const int nlev = A.sum_levels ? 1 : M + 1;
                        for (int lv = 0; lv < nlev; ++lv)
                            for (int r = wave * 4 + r4; r < 64; r += NW * 4) {
                                const int64_t tr = tb * int64_t(64) + r;
                                if (tr < A.Tn && cc <= col) {
                                    double v;
                                    if (A.sum_levels) {
                                        v = tile[(0 * 64 + r) * TS + cc];
#pragma unroll
                                        for (int ww = 1; ww < NW; ++ww) v += tile[(ww * 64 + r) * TS + cc];
                                    } else {
                                        v = tile[(lv * 64 + r) * TS + cc];
                                    }
                                    out[(int64_t(lv) * A.Tn + tr) * A.N + nb + cc] = v;
                                }
                            }
                        if (n + 1 == n_end) it = __builtin_amdgcn_readfirstlane(sh_item[0]);
                        __syncthreads();                                  // the tile (and the item slot) is free again
                    }
                }
            }
        }
    };

    if constexpr (NW == 1) {
        TvsTileWave<M, NW, D, INCR, KIND, tvs_level_mask(M, 1, 0)> W;
        run_wave(W);
    } else {
        if (wave == 0) {
            TvsTileWave<M, NW, D, INCR, KIND, tvs_level_mask(M, NW, 0)> W;
            run_wave(W);
        } else if (NW == 2 || wave == 1) {
            TvsTileWave<M, NW, D, INCR, KIND, tvs_level_mask(M, NW, 1)> W;
            run_wave(W);
        } else {
            TvsTileWave<M, NW, D, INCR, KIND, tvs_level_mask(M, NW, NW > 2 ? 2 : 0)> W;
            run_wave(W);
        }
    }
}

// Prepared inducing tensors in the tensor-lane layout.  In: Z (lt, T, E_in, d_eff).  Out:
//   ZL[((k * E + e) * D + fe) * Tpad + t] = pre * z~ (zero for fe >= d_eff),   ZN[(k * E + e) * Tpad + t] = |pre * z~|^2   (zero for t >= T)
// with z~ the scaled component (kernels.py:367-398) and, for collapse (linear kernel, E_in = 2, E = 1), the difference of the
// component's two points (kernels.py:329-330 applied before the inner product, which is linear in it).
static __global__ void prep_tensors_tile_kernel(const double* __restrict__ Z, int lt, int64_t Tn, int64_t Tpad, int E_in, int collapse,
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

// Records of the sequences: rec[n][tau * D + fe] = pre * x~[n][tau][fe]  (increments == 1: x~[tau] - x~[tau-1], row 0
// zero; columns fe >= d_eff zero), then rec[n][L * D + tau] = |pre * x~[n][tau]|^2; the tail up to rec_elems is zero-filled here.
static __global__ void prep_seq_tile_records_kernel(const double* __restrict__ X, int64_t N, int L, ScaleParams P, double pre,
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

// Rows of the sequences for the forward tile kernel (TvsTileArgs::XR): row[n * L + tau][fe] = pre * x~[n][tau][fe]  (increments == 1:
// x~[tau] - x~[tau-1], row 0 of a sequence zero; columns fe >= d_eff zero), row[.][D] = |pre * x~[n][tau]|^2, the rest of the RS doubles and the one
// extra row behind the last sequence zero.
static __global__ void prep_seq_tile_rows_kernel(const double* __restrict__ X, int64_t N, int L, ScaleParams P, double pre,
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

}  // namespace gpsig
