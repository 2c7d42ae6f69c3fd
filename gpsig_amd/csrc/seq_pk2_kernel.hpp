// seq_pk2_kernel.hpp -- float32 sequence-vs-sequence Gram with TWO y-side sequences per pair group, packed.
//
// Same task structure, LDS ring, lane skew and epilogue as seq_gram_kernel.hpp (first-order algorithm,
// gpsig/signature_algs.py:8-35; SignatureKernel._K_seq + the epilogue of K, gpsig/kernels.py:208-237, :430-476), but every
// lane carries the lattice columns of two consecutive y sequences (A, B) against the one streamed x: all per-lane state is
// a pair of floats in one 64-bit register pair, and the whole step -- inner products, double increment, level recursion --
// runs on the packed float32 instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32; the x row is the broadcast operand).
// What that buys on gfx950, measured (profiles/r02_ab_variants.txt): NOT the factor two the instruction count suggests.
// A SIMD issues a scalar float32 instruction for 64 lanes in 2 cycles and a packed one in 4 (MI355X_MICROARCH.md,
// "Wave scheduling"; the 118 TFLOP/s "v_fma_f32" line of profiles/r01_microbench_gfx950.txt was compiled to v_pk_fma_f32),
// so packing halves the instructions without shortening their issue time.  The RBF kernel still gains 14 % at BASELINE
// configs[4] (41.5 -> 35.8 ms: no register copies between steps, one exp argument per packed accumulator); the linear kernel
// loses (28.2 -> 33.5 ms: two columns per lane give its inner products too little independent work), so the planner uses this
// kernel for the RBF family only.  Packing two y's (rather than two x's) leaves the x records, the ring and its LDS
// footprint exactly as they are.
//
// Built for the shapes in seq_pk2_inst.hip: exact num_levels, MODE_INC (linear kernel; off by default, see above) and
// MODE_PT_DIFF with the RBF kernel.  RBF records are prescaled by sqrt(log2 e) with -|x'|^2/2 in the spare column of every row (the same
// arrangement as the float64 kernels, seq_step_rbf_prescaled in seq_core.hpp), so kappa = v_exp_f32(<x',y'> + hx + hy).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "seq_args.hpp"
#include "seq_core.hpp"
#include "seq_gram_kernel.hpp"

namespace gpsig {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr float PK2_RBF_PRESCALE = 1.2011224087864498f;      // sqrt(log2 e): <x', y'> - |x'|^2/2 - |y'|^2/2 = log2 kappa

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 pk_splat(float x) { return f2{x, x}; }
template <int G>
__device__ __forceinline__ f2 pk_shr1(f2 v) { return f2{shr1<G>(v.x), shr1<G>(v.y)}; }

template <int C, int D, int M, int MODE>
struct SeqLanePk2 {
    static constexpr int NQ = M > 1 ? M - 1 : 1;
    f2 y[C][D];       // record rows of (y_A, y_B) owned by this lane
    f2 hy[C];         // RBF: -|y'|^2 / 2
    f2 q[NQ][C];      // Q_m, m = 1 .. M-1
    f2 qg[NQ];        // ghost column (SeqLane::qg)
    f2 s[M];          // end-of-chunk row prefixes: the hand-over words
    f2 ktop;
    f2 eprev[C];      // point mode: kappa(x_prev, y_r) - kappa(x_prev, y_{r-1})
    f2 klast;         // point mode: kappa(x, last owned column) of the last processed row (the right neighbour's k_left)

    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int m = 0; m < NQ; ++m) {
#pragma unroll
            for (int r = 0; r < C; ++r) q[m][r] = pk_splat(0.f);
            qg[m] = pk_splat(0.f);
        }
        ktop = pk_splat(0.f);
    }
    __device__ __forceinline__ void init() {
        reset();
#pragma unroll
        for (int m = 0; m < M; ++m) s[m] = pk_splat(0.f);
#pragma unroll
        for (int r = 0; r < C; ++r) { eprev[r] = pk_splat(0.f); hy[r] = pk_splat(0.f); }
        klast = pk_splat(0.f);
    }
    __device__ __forceinline__ f2 level(int m) const {         // K_m as seen by the last lane of the group
        f2 v = ktop;
#pragma unroll
        for (int k = 0; k < NQ; ++k)
            if (k == m - 1 && m < M) v = q[k][C - 1];
        return v;
    }
};

// the recursion of seq_core.hpp's seq_level on packed pairs; MI descends so that level m+1 reads Q_m before it moves
template <int MI, int G, int C, int D, int M, int MODE>
__device__ __forceinline__ void pk2_level(SeqLanePk2<C, D, M, MODE>& L, const f2 (&dm)[C]) {
    const f2 cin = pk_shr1<G>(L.s[MI]);
    f2 sm = cin;
    if constexpr (MI == M - 1) {
        if constexpr (MI == 0) {
#pragma unroll
            for (int r = 0; r < C; ++r) sm += dm[r];
        } else {
            sm = pk_fma(dm[0], L.qg[MI - 1], sm);
#pragma unroll
            for (int r = 1; r < C; ++r) sm = pk_fma(dm[r], L.q[MI - 1][r - 1], sm);
        }
        L.ktop += sm;
    } else {
        if constexpr (MI == 0) {
#pragma unroll
            for (int r = 0; r < C; ++r) { sm += dm[r]; L.q[0][r] += sm; }
        } else {
            sm = pk_fma(dm[0], L.qg[MI - 1], sm);
            L.q[MI][0] += sm;
#pragma unroll
            for (int r = 1; r < C; ++r) { sm = pk_fma(dm[r], L.q[MI - 1][r - 1], sm); L.q[MI][r] += sm; }
        }
        L.qg[MI] += cin;
    }
    L.s[MI] = sm;
    if constexpr (MI > 0) pk2_level<MI - 1, G>(L, dm);
}

template <int G, int C, int D, int M, int MODE>
__device__ __forceinline__ void pk2_step(SeqLanePk2<C, D, M, MODE>& L, const float (&xr)[D], float hx, bool dummy, int rlo, int rhi) {
    f2 dm[C];
    if constexpr (MODE == MODE_INC) {
#pragma unroll
        for (int r = 0; r < C; ++r) {
            f2 acc = pk_splat(xr[0]) * L.y[r][0];
#pragma unroll
            for (int f = 1; f < D; ++f) acc = pk_fma(pk_splat(xr[f]), L.y[r][f], acc);
            dm[r] = acc;
        }
    } else {
        f2 knew[C];
#pragma unroll
        for (int r = 0; r < C; ++r) {
            f2 acc = L.hy[r] + pk_splat(hx);
#pragma unroll
            for (int f = 0; f < D; ++f) acc = pk_fma(pk_splat(xr[f]), L.y[r][f], acc);
            knew[r] = f2{__builtin_amdgcn_exp2f(acc.x), __builtin_amdgcn_exp2f(acc.y)};
        }
        const f2 kl = pk_shr1<G>(L.klast);      // the left neighbour's last column at this x row (it was there one step ago)
#pragma unroll
        for (int r = 0; r < C; ++r) {
            const f2 e = knew[r] - (r == 0 ? kl : knew[r == 0 ? 0 : r - 1]);
            dm[r] = (dummy || r < rlo || r >= rhi) ? pk_splat(0.f) : e - L.eprev[r];     // signature_algs.py:26
            L.eprev[r] = e;
        }
        L.klast = knew[C - 1];
    }
    pk2_level<M - 1, G>(L, dm);
}

template <int G, int C, int D, int M, int MODE>
__global__ __launch_bounds__(64, 2) void seq_pk2_kernel(const SeqGramArgs A) {
    static_assert(G == 16 || G == 64, "pair group is a DPP row or the whole wave");
    static_assert(D % 4 == 0, "record rows are read with 16-byte LDS loads");
    using Lane = SeqLanePk2<C, D, M, MODE>;
    typedef float vecT __attribute__((ext_vector_type(4)));
    constexpr bool RBF = MODE != MODE_INC;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* const zero_row = reinterpret_cast<float*>(smem_raw);
    float* const ring = zero_row + A.RS;

    const int lane = threadIdx.x;
    const int lam = lane & (G - 1);
    const int grp = lane / G;
    const SeqTask tk = A.tasks[blockIdx.x];
    const int R1 = A.R1, RS = A.RS, nslot = A.nslot, nx = tk.nx;
    const float* const xrec = static_cast<const float*>(A.xrec);
    const float* const yrec = static_cast<const float*>(A.yrec);

    if (lane < RS) zero_row[lane] = 0.f;

    // ---- y side: C record rows of sequences jA, jA + 1 ----------------------------------------------
    const int64_t jA = int64_t(tk.y0) + 2 * grp, jB = jA + 1;
    const bool validA = jA < A.N2, validB = jB < A.N2;
    Lane L;
    L.init();
#pragma unroll
    for (int r = 0; r < C; ++r) {
        const int row = C * lam + r;
        const bool okA = validA && row < A.R2, okB = validB && row < A.R2;
        const float* sa = yrec + (okA ? jA * A.yrec_stride + int64_t(row) * RS : 0);
        const float* sb = yrec + (okB ? jB * A.yrec_stride + int64_t(row) * RS : 0);
        f2 ys = pk_splat(0.f);
#pragma unroll
        for (int f = 0; f < D; f += 4) {
            const vecT va = okA ? *reinterpret_cast<const vecT*>(sa + f) : vecT(0.f);
            const vecT vb = okB ? *reinterpret_cast<const vecT*>(sb + f) : vecT(0.f);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                L.y[r][f + e] = f2{va[e], vb[e]};
                ys = pk_fma(L.y[r][f + e], L.y[r][f + e], ys);
            }
        }
        L.hy[r] = -0.5f * ys;
    }
    const int rlo = (lam == 0) ? 1 : 0;
    int rhi = A.R2 - C * lam;
    rhi = rhi < 0 ? 0 : (rhi > C ? C : rhi);

    // ---- x side staging (as seq_gram_kernel) -----------------------------------------------------------
    auto stage = [&](int p, int slot) {
        int64_t i = int64_t(tk.x0) + p;
        if (i >= A.N1) i -= A.N1;
        const float* src = xrec + i * A.xrec_stride;
        float* dst = ring + int64_t(slot) * A.slot_elems;
        const int pieces = A.slot_elems / 4;              // multiple of 64
        if (A.use_glds) {
            for (int c = 0; c < pieces; c += 64)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + int64_t(c + lane) * 4),
                                                 (__attribute__((address_space(3))) void*)(dst + int64_t(c) * 4), 16, 0, 0);
        } else {
            for (int c = lane; c < pieces; c += 64) reinterpret_cast<vecT*>(dst)[c] = reinterpret_cast<const vecT*>(src)[c];
        }
    };
    stage(0, 0);

    LaneCtl ctl;
    ctl.init(lam, RS);
    const int ring_elems = nslot * A.slot_elems;
    const int nsteps = nx * R1 + G;
    int a_u = 0, k_u = 0, slot_next = 1 % nslot;
    if (A.use_glds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    struct Half {                                         // one of the two packed problems, for the shared epilogue
        const Lane& L;
        int h;
        __device__ __forceinline__ float level_value(int m, int) const { const f2 v = L.level(m); return h ? v.y : v.x; }
    };

    auto one_step = [&]() {
        if (a_u == A.issue_at && k_u + 1 < nx) {
            stage(k_u + 1, slot_next);
            if (++slot_next == nslot) slot_next = 0;
        }
        if (++a_u == R1) { a_u = 0; ++k_u; }

        if (ctl.begin_step(nx, R1, RS, A.slot_elems, ring_elems)) {   // the pair that just finished is complete in the group's last lane
            if (lam == G - 1 && ctl.p >= 1) {
                int64_t i = int64_t(tk.x0) + (ctl.p - 1);
                if (i >= A.N1) i -= A.N1;
                float* const out = static_cast<float*>(A.out);
                if (validA) seq_emit<float>(Half{L, 0}, A, i, jA, M, [&](int64_t off, float v) { out[off] = v; });
                if (validB) seq_emit<float>(Half{L, 1}, A, i, jB, M, [&](int64_t off, float v) { out[off] = v; });
            }
            L.reset();
        }

        float xr[D];
        const float* rowp = zero_row + ctl.rowoff;
#pragma unroll
        for (int f = 0; f < D; f += 4) {
            const vecT v = *reinterpret_cast<const vecT*>(rowp + f);
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[f + e] = v[e];
        }
        float hx = 0.f;
        if constexpr (RBF) hx = rowp[D];                  // -|x'|^2 / 2, the record row's spare column
        const bool dummy = ctl.row0;
        // if lane 0 opens a new x at the next step, its record (requested issue_at steps into this x) must have landed
        if (a_u == 0 && A.use_glds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pk2_step<G>(L, xr, hx, dummy, rlo, rhi);
        ctl.end_step();
    };
    // two steps per trip (an odd count is rounded up: the extra step finds every lane idle).  Requesting the next step's row
    // one step ahead and splitting the inner products into two accumulation chains were both measured and lost (BASELINE
    // configs[4]: 35.8 -> 39.2 ms): profiles/r02_ab_variants.txt.
    for (int t = 0; t < nsteps; t += 2) {
        one_step();
        one_step();
    }
}

template <int G, int C, int D, int M, int MODE>
hipError_t seq_pk2_launch(const SeqGramArgs& A, int ntasks, size_t lds_bytes, hipStream_t stream) {
    if (ntasks <= 0) return hipSuccess;
    auto kern = seq_pk2_kernel<G, C, D, M, MODE>;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(ntasks), dim3(64), lds_bytes, stream, A);
    return hipGetLastError();
}

}  // namespace gpsig
