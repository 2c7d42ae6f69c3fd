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

// Entries of the exp table (fast_exp.hpp): 64 (degree-5 tail), 256 (degree 4), 1024 / 2048 (degree 3, one instruction fewer per exp).
// Same box, alternating processes: 64 -> 256 entries: Kzx RBF 3.20 -> 3.07 ms, with increments 6.49 -> 6.17 ms (profiles/r02_ab_exp256.txt);
// 256 -> 1024 -> 2048 entries: 2.93 -> 3.34 -> 3.70 ms WITHOUT increments (8 / 16 KB more LDS per workgroup cost the stream-bound case
// its occupancy), 6.07 -> 5.88 -> 5.89 ms WITH increments (ten exps per step and wave: the instruction counts) -- profiles/r02_ab_exptab.txt.
// Hence by kernel: incremental tensors take 1024 entries, the others 256.  TVS_EXPTAB (64 ... 2048) forces one size for A/B builds.
#ifdef TVS_EXPTAB
constexpr int tvs_etab_n(bool) { return TVS_EXPTAB; }
#else
constexpr int tvs_etab_n(bool two_points) { return two_points ? 1024 : 256; }
#endif
constexpr double tvs_rbf_prescale(bool two_points) {
    return tvs_etab_n(two_points) == 64 ? EXP_PRESCALE : (tvs_etab_n(two_points) == 256 ? EXP_PRESCALE256 : (tvs_etab_n(two_points) == 1024 ? 4.0 * EXP_PRESCALE : 0x1.b2da4e9808a53p+5));
}
template <int NTAB>
__device__ __forceinline__ double tvs_exp2(double t, const double* etab) {
    if constexpr (NTAB == 64) return kexp2_tab(t, etab);
    else if constexpr (NTAB == 256) return kexp2_tab256(t, etab);
    else return kexp2_tabn<NTAB>(t, etab);
}
constexpr int TVS_TILE_S = 16;           // sequences per output flush: 16 doubles = one 128-byte line per tensor row
constexpr int TVS_REC_ALIGN = 128;       // record length granule in elements of double: 64 lanes x 16 bytes of LDS-DMA

struct TvsTileArgs {
    const void* XR;     // (N, rec_elems) records: L rows of D prepared values (D = the kernel's feature width, zero beyond d),
                        // then L squared norms of those rows' POINTS
    const void* ZL;     // (lt, E, D, Tpad) prepared tensor components, tensor index fastest, zero rows beyond d
    const void* ZN;     // (lt, E, Tpad) squared norms of the prepared components
    int64_t N, Tn, Tpad;
    int32_t L, d, kind, difference, M;
    int32_t run;        // sequences per workgroup
    int32_t rec_elems;  // multiple of TVS_REC_ALIGN
    double p0, p1;
    const void* fx;     // (N, M+1) per-sequence factors or NULL
    const double* w;    // (M+1) level weights or NULL
    void* out;          // (T, N) or (M+1, T, N)
    int32_t sum_levels;
    double* aux;        // optional (N, lt, Tpad): the totals of EVERY chain, u_{j+1} of component k = i(i-1)/2 + j at [n][k][t] -- what the
                        // reverse pass (tvs_grad_tile_kernel.hpp) would otherwise rebuild with a forward sweep of its own
};

// LDS bytes of one workgroup
inline size_t tvs_tile_lds_bytes(int M, int NW, int rec_elems, bool sum_levels, bool two_points) {
    const size_t slots = sum_levels ? size_t(NW) : size_t(M + 1);
    return sizeof(double) * (tvs_etab_n(two_points) + 2 * size_t(rec_elems) + slots * 64 * (TVS_TILE_S + 1));
}

template <int M, int NW, int D, bool INCR, int KIND, int MASK>
struct TvsTileWave {
    static constexpr int E = (INCR && KIND != BASE_LINEAR) ? 2 : 1;        // linear + increments arrives collapsed
    static constexpr int NC = tvs_mask_comps(MASK);

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

    // k1[c]: the component kernels at time tau (kernels.py:323-330), before the difference along time
    __device__ __forceinline__ void eval(const TvsTileArgs& A, const double* __restrict__ rec, const double* __restrict__ etab,
                                         int tau, double (&k1)[NC]) const {
        double x[D];
#pragma unroll
        for (int f = 0; f < D; ++f) x[f] = rec[tau * D + f];
        double kv[NC * E], a2[NC * E];
        if constexpr (KIND == BASE_RBF) {
            const double hx = -0.5 * rec[A.L * D + tau];
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    double t = zn[c][e] + hx;
#pragma unroll
                    for (int f = 0; f < D; ++f) t = fma(z[c][e][f], x[f], t);
                    kv[c * E + e] = tvs_exp2<tvs_etab_n(E == 2)>(t, etab);
                }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    double ip = z[c][e][0] * x[0];
#pragma unroll
                    for (int f = 1; f < D; ++f) ip = fma(z[c][e][f], x[f], ip);
                    kv[c * E + e] = ip;
                    a2[c * E + e] = zn[c][e];
                }
            if constexpr (KIND != BASE_LINEAR) base_eval_n<double, NC * E>(A.kind, kv, a2, rec[A.L * D + tau], A.p0, A.p1);
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

    __device__ __forceinline__ void sweep(const TvsTileArgs& A, const double* __restrict__ rec, const double* __restrict__ etab) {
        const int L = A.L;
        double ka[NC], kb[NC], dm[NC];
        if (!A.difference) {
            for (int tau = 0; tau < L; ++tau) {
                eval(A, rec, etab, tau, ka);
                chains(ka);
            }
        } else if constexpr (KIND == BASE_LINEAR) {               // rows are increments already (row 0 unused)
            for (int tau = 1; tau < L; ++tau) {
                eval(A, rec, etab, tau, ka);
                chains(ka);
            }
        } else {                                                  // signature_algs.py:114: difference along time
            eval(A, rec, etab, 0, ka);
            int tau = 1;
            for (; tau + 1 < L; tau += 2) {                       // two steps per trip: the previous values alternate registers
                eval(A, rec, etab, tau, kb);
#pragma unroll
                for (int c = 0; c < NC; ++c) dm[c] = kb[c] - ka[c];
                chains(dm);
                eval(A, rec, etab, tau + 1, ka);
#pragma unroll
                for (int c = 0; c < NC; ++c) dm[c] = ka[c] - kb[c];
                chains(dm);
            }
            if (tau < L) {
                eval(A, rec, etab, tau, kb);
#pragma unroll
                for (int c = 0; c < NC; ++c) dm[c] = kb[c] - ka[c];
                chains(dm);
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

template <int M, int NW, int D, bool INCR, int KIND>
__global__ __launch_bounds__(NW * 64, 2) void tvs_tile_kernel(const TvsTileArgs A) {
    constexpr int TS = TVS_TILE_S + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char tvs_tile_smem[];
    double* const etab = reinterpret_cast<double*>(tvs_tile_smem);
    constexpr int NTAB = tvs_etab_n(INCR && KIND != BASE_LINEAR);
    double* const recs = etab + NTAB;                              // 2 x rec_elems
    double* const tile = recs + 2 * A.rec_elems;                  // [slots][64][TS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t t = blockIdx.x * int64_t(64) + lane;            // < Tpad
    const int64_t n_begin = blockIdx.y * int64_t(A.run);
    const int64_t n_end = (n_begin + A.run < A.N) ? n_begin + A.run : A.N;
    const double* __restrict__ XR = static_cast<const double*>(A.XR);
    const double* __restrict__ fx = static_cast<const double*>(A.fx);
    double* __restrict__ out = static_cast<double*>(A.out);

    if constexpr (KIND != BASE_LINEAR) {
        if constexpr (NTAB == 64) exp_tab_fill(etab, tid, NW * 64);
        else if constexpr (NTAB == 256) exp_tab256_fill(etab, tid, NW * 64);
        else exp_tabn_fill<NTAB>(etab, tid, NW * 64);
    }

    // records arrive by LDS-DMA: 64 lanes x 16 bytes per instruction, the waves take alternate kilobytes
    auto stage = [&](int64_t n, int buf) {
        const double* src = XR + n * int64_t(A.rec_elems);
        double* dst = recs + buf * A.rec_elems;
        for (int c = wave * 128; c < A.rec_elems; c += NW * 128)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c + lane * 2),
                                             (__attribute__((address_space(3))) void*)(dst + c), 16, 0, 0);
    };
    if (n_begin < n_end) stage(n_begin, 0);

    auto run_wave = [&](auto& W) {
        W.load(A, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int64_t n = n_begin; n < n_end; ++n) {
            const int idx = int(n - n_begin), buf = idx & 1, col = idx % TVS_TILE_S;
            if (n + 1 < n_end) stage(n + 1, buf ^ 1);
            double fac[M + 1];
#pragma unroll
            for (int i = 0; i <= M; ++i) {
                double f = fx ? fx[n * (M + 1) + i] : 1.0;
                if (A.w) f *= A.w[i];
                fac[i] = f;
            }
            W.sweep(A, recs + buf * A.rec_elems, etab);
            if (A.aux) W.store_aux(A, n, t);
            W.emit(A, fac, tile, wave, lane, col, wave == 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next record has landed
            __syncthreads();                                      // ... for every wave, and this record is no longer read
            if (col == TVS_TILE_S - 1 || n + 1 == n_end) {
                // flush: 4 tensor rows x 16 sequences per store instruction
                const int64_t nb = n - col;                       // first sequence of the tile
                const int r4 = lane >> 4, cc = lane & 15;
                const int nlev = A.sum_levels ? 1 : M + 1;
                for (int lv = 0; lv < nlev; ++lv)
                    for (int r = wave * 4 + r4; r < 64; r += NW * 4) {
                        const int64_t tr = blockIdx.x * int64_t(64) + r;
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
                __syncthreads();                                  // the tile is free again
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
                                         double pre, ScaleParams P, int D, double* __restrict__ ZL, double* __restrict__ ZN) {
    const int d_eff = P.d_eff();
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

}  // namespace gpsig
