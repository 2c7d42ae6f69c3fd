// seq_pk2_kernel.hpp -- float32 sequence-vs-sequence Gram with TWO y-side sequences per pair group, packed.
//
// Same task structure, LDS ring, lane skew and epilogue as seq_gram_kernel.hpp (first-order algorithm,
// gpsig/signature_algs.py:8-35; SignatureKernel._K_seq + the epilogue of K, gpsig/kernels.py:208-237, :430-476), but every
// lane carries the lattice columns of two consecutive y sequences (A, B) against the one streamed x: all per-lane state is
// a pair of floats in one 64-bit register pair, and the whole step -- inner products, double increment, level recursion --
// runs on the packed float32 instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32; the x row is the broadcast operand).
// Packing two y's (rather than two x's) leaves the x records, the ring and its LDS footprint exactly as they are.
//
// What it buys on gfx950, measured at BASELINE configs[4] (profiles/r02_bench_c5_variants.txt; RBF / linear, ms per Gram):
//     one-sequence kernels <float,16,8,16,6> (round 1)          41.6 / 28.0
//     this kernel, V = float (one y, scalar instructions)        57.7 / 51.6      4 waves on one ring: 38.8 / 31.9
//     this kernel, V = f2    (two y's, packed instructions)      38.6 / 33.5      4 waves on one ring: 30.5 / 26.2
// A SIMD issues a scalar float32 instruction for 64 lanes in 2 cycles and a packed one in 4 (MI355X_MICROARCH.md, "Wave
// scheduling"; the 118 TFLOP/s "v_fma_f32" line of profiles/r01_microbench_gfx950.txt was compiled to v_pk_fma_f32), so
// packing does not double the issue-bound rate.  These kernels are not issue-bound, though: the unpacked round-1 kernel runs
// at 56 % of its 2-cycle issue floor, on dependent-instruction latency at two wavefronts per SIMD.  Halving the number of
// dependent instructions per cell is worth a factor 0.65-0.8 at equal structure; the other factor 0.8 comes from occupancy
// (template parameter W below): one ring of x records per workgroup of four waves instead of 20 KB of LDS per wave.
//
// Built for the shapes in seq_pk2_inst.hip: exact num_levels, MODE_INC (linear kernel; the planner takes it for launches
// large enough for the four-wave workgroups only) and MODE_PT_DIFF with the RBF kernel.  RBF records are prescaled by sqrt(log2 e) with -|x'|^2/2 in the spare column of every row (the same
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

// V = f2: two y sequences per pair group (packed instructions);  V = float: one (the same code, scalar instructions)
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float pk_fma(float a, float b, float c) { return fmaf(a, b, c); }
template <typename V> __device__ __forceinline__ V pk_splat(float x);
template <> __device__ __forceinline__ f2 pk_splat<f2>(float x) { return f2{x, x}; }
template <> __device__ __forceinline__ float pk_splat<float>(float x) { return x; }
template <int G>
__device__ __forceinline__ f2 pk_shr1(f2 v) { return f2{shr1<G>(v.x), shr1<G>(v.y)}; }
template <int G>
__device__ __forceinline__ float pk_shr1(float v) { return shr1<G>(v); }
__device__ __forceinline__ f2 pk_exp2(f2 a) { return f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
__device__ __forceinline__ float pk_exp2(float a) { return __builtin_amdgcn_exp2f(a); }
__device__ __forceinline__ f2 pk_pair(float a, float b, f2) { return f2{a, b}; }
__device__ __forceinline__ float pk_pair(float a, float, float) { return a; }
__device__ __forceinline__ float pk_half(f2 v, int h) { return h ? v.y : v.x; }
__device__ __forceinline__ float pk_half(float v, int) { return v; }
template <typename V> struct PkWidth { static constexpr int N = 1; };
template <> struct PkWidth<f2> { static constexpr int N = 2; };

template <typename V, int C, int D, int M, int MODE>
struct SeqLanePk2 {
    static constexpr int NQ = M > 1 ? M - 1 : 1;
    V y[C][D];        // record rows of (y_A, y_B) owned by this lane
    V hy[C];          // RBF: -|y'|^2 / 2
    V q[NQ][C];       // Q_m, m = 1 .. M-1
    V qg[NQ];         // ghost column (SeqLane::qg)
    V s[M];           // end-of-chunk row prefixes: the hand-over words
    V ktop;
    V eprev[C];       // point mode: kappa(x_prev, y_r) - kappa(x_prev, y_{r-1})
    V klast;          // point mode: kappa(x, last owned column) of the last processed row (the right neighbour's k_left)
    V keep;           // SeqLane::keep: 1, or 0 at the step that opens a new pair

    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int m = 0; m < NQ; ++m) {
#pragma unroll
            for (int r = 0; r < C; ++r) q[m][r] = pk_splat<V>(0.f);
            qg[m] = pk_splat<V>(0.f);
        }
        ktop = pk_splat<V>(0.f);
    }
    __device__ __forceinline__ void init() {
        reset();
#pragma unroll
        for (int m = 0; m < M; ++m) s[m] = pk_splat<V>(0.f);
#pragma unroll
        for (int r = 0; r < C; ++r) { eprev[r] = pk_splat<V>(0.f); hy[r] = pk_splat<V>(0.f); }
        klast = pk_splat<V>(0.f);
        keep = pk_splat<V>(1.f);
    }
    __device__ __forceinline__ V level(int m) const {         // K_m as seen by the last lane of the group
        V v = ktop;
#pragma unroll
        for (int k = 0; k < NQ; ++k)
            if (k == m - 1 && m < M) v = q[k][C - 1];
        return v;
    }
};

// the recursion of seq_core.hpp's seq_level on packed pairs; MI descends so that level m+1 reads Q_m before it moves
template <int MI, int G, typename V, int C, int D, int M, int MODE>
__device__ __forceinline__ void pk2_level(SeqLanePk2<V, C, D, M, MODE>& L, const V (&dm)[C]) {
    const V cin = pk_shr1<G>(L.s[MI]);
    V sm = cin;
    if constexpr (MI == M - 1) {
        if constexpr (MI == 0) {
#pragma unroll
            for (int r = 0; r < C; ++r) sm += dm[r];
        } else {
            sm = pk_fma(dm[0], L.qg[MI - 1], sm);
#pragma unroll
            for (int r = 1; r < C; ++r) sm = pk_fma(dm[r], L.q[MI - 1][r - 1], sm);
        }
        L.ktop = pk_fma(L.ktop, L.keep, sm);
    } else {
        if constexpr (MI == 0) {
#pragma unroll
            for (int r = 0; r < C; ++r) { sm += dm[r]; L.q[0][r] = pk_fma(L.q[0][r], L.keep, sm); }
        } else {
            sm = pk_fma(dm[0], L.qg[MI - 1], sm);
            L.q[MI][0] = pk_fma(L.q[MI][0], L.keep, sm);
#pragma unroll
            for (int r = 1; r < C; ++r) { sm = pk_fma(dm[r], L.q[MI - 1][r - 1], sm); L.q[MI][r] = pk_fma(L.q[MI][r], L.keep, sm); }
        }
        L.qg[MI] = pk_fma(L.qg[MI], L.keep, cin);
    }
    L.s[MI] = sm;
    if constexpr (MI > 0) pk2_level<MI - 1, G>(L, dm);
}

template <int G, typename V, int C, int D, int M, int MODE>
__device__ __forceinline__ void pk2_step(SeqLanePk2<V, C, D, M, MODE>& L, const float (&xr)[D], float hx, bool dummy, int rlo, int rhi) {
    V dm[C];
    if constexpr (MODE == MODE_INC) {
#pragma unroll
        for (int r = 0; r < C; ++r) {
            V acc = pk_splat<V>(xr[0]) * L.y[r][0];
#pragma unroll
            for (int f = 1; f < D; ++f) acc = pk_fma(pk_splat<V>(xr[f]), L.y[r][f], acc);
            dm[r] = acc;
        }
    } else {
        V knew[C];
#pragma unroll
        for (int r = 0; r < C; ++r) {
            V acc = L.hy[r] + pk_splat<V>(hx);
#pragma unroll
            for (int f = 0; f < D; ++f) acc = pk_fma(pk_splat<V>(xr[f]), L.y[r][f], acc);
            knew[r] = pk_exp2(acc);
        }
        const V kl = pk_shr1<G>(L.klast);      // the left neighbour's last column at this x row (it was there one step ago)
#pragma unroll
        for (int r = 0; r < C; ++r) {
            const V e = knew[r] - (r == 0 ? kl : knew[r == 0 ? 0 : r - 1]);
            dm[r] = (dummy || r < rlo || r >= rhi) ? pk_splat<V>(0.f) : e - L.eprev[r];     // signature_algs.py:26
            L.eprev[r] = e;
        }
        L.klast = knew[C - 1];
    }
    pk2_level<M - 1, G>(L, dm);
}

// V: f2 (two y sequences per pair group, packed instructions) or float (one).  W: wavefronts per workgroup.  The W waves of a
// workgroup take W consecutive y blocks of the task and share ONE ring of x records: an x record is staged once for W times
// as many pairs, and the ring's LDS is paid once per workgroup instead of once per wave (20 KB per wave capped a CU at 8
// wavefronts at BASELINE configs[4]).  The waves meet at two barriers per x: before a ring slot is overwritten (every lane of
// every wave has left it: seq_ring's issue_at argument, per wave) and after its DMA pieces have landed.
template <typename V, int W, int G, int C, int D, int M, int MODE>
__global__ __launch_bounds__(W * 64, W == 1 ? 2 : 1) void seq_pk2_kernel(const SeqGramArgs A) {
    static_assert(G == 16 || G == 64, "pair group is a DPP row or the whole wave");
    static_assert(D % 4 == 0, "record rows are read with 16-byte LDS loads");
    using Lane = SeqLanePk2<V, C, D, M, MODE>;
    typedef float vecT __attribute__((ext_vector_type(4)));
    constexpr bool RBF = MODE != MODE_INC;
    constexpr int NY = PkWidth<V>::N;                     // y sequences per pair group

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* const zero_row = reinterpret_cast<float*>(smem_raw);
    float* const ring = zero_row + A.RS;

    const int lane = threadIdx.x & 63;
    const int wave = W == 1 ? 0 : __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
    const int lam = lane & (G - 1);
    const int grp = lane / G;
    const SeqTask tk = A.tasks[blockIdx.x];
    const int R1 = A.R1, RS = A.RS, nslot = A.nslot, nx = tk.nx;
    const float* const xrec = static_cast<const float*>(A.xrec);
    const float* const yrec = static_cast<const float*>(A.yrec);
    auto sync = [&]() { if constexpr (W > 1) __syncthreads(); };

    if (wave == 0 && lane < RS) zero_row[lane] = 0.f;

    // ---- y side: C record rows of sequences jA (and jA + 1) ------------------------------------------
    const int64_t jA = int64_t(tk.y0) + (int64_t(wave) * (64 / G) + grp) * NY, jB = jA + 1;
    const bool validA = jA < A.N2, validB = NY == 2 && jB < A.N2;
    Lane L;
    L.init();
#pragma unroll
    for (int r = 0; r < C; ++r) {
        const int row = C * lam + r;
        const bool okA = validA && row < A.R2, okB = validB && row < A.R2;
        const float* sa = yrec + (okA ? jA * A.yrec_stride + int64_t(row) * RS : 0);
        const float* sb = yrec + (okB ? jB * A.yrec_stride + int64_t(row) * RS : 0);
        V ys = pk_splat<V>(0.f);
#pragma unroll
        for (int f = 0; f < D; f += 4) {
            const vecT va = okA ? *reinterpret_cast<const vecT*>(sa + f) : vecT(0.f);
            vecT vb = vecT(0.f);
            if constexpr (NY == 2) vb = okB ? *reinterpret_cast<const vecT*>(sb + f) : vecT(0.f);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                L.y[r][f + e] = pk_pair(va[e], vb[e], V());
                ys = pk_fma(L.y[r][f + e], L.y[r][f + e], ys);
            }
        }
        L.hy[r] = -0.5f * ys;
    }
    const int rlo = (lam == 0) ? 1 : 0;
    int rhi = A.R2 - C * lam;
    rhi = rhi < 0 ? 0 : (rhi > C ? C : rhi);

    // ---- x side staging (as seq_gram_kernel; the waves of the workgroup take alternate kilobytes) -----------
    auto stage = [&](int p, int slot) {
        int64_t i = int64_t(tk.x0) + p;
        if (i >= A.N1) i -= A.N1;
        const float* src = xrec + i * A.xrec_stride;
        float* dst = ring + int64_t(slot) * A.slot_elems;
        const int pieces = A.slot_elems / 4;              // multiple of 64
        if (A.use_glds) {
            for (int c = wave * 64; c < pieces; c += W * 64)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + int64_t(c + lane) * 4),
                                                 (__attribute__((address_space(3))) void*)(dst + int64_t(c) * 4), 16, 0, 0);
        } else {
            for (int c = wave * 64 + lane; c < pieces; c += W * 64) reinterpret_cast<vecT*>(dst)[c] = reinterpret_cast<const vecT*>(src)[c];
        }
    };
    stage(0, 0);

    LaneCtl ctl;
    ctl.init(lam, RS);
    const int ring_elems = nslot * A.slot_elems;
    const int nsteps = nx * R1 + G;
    int a_u = 0, k_u = 0, slot_next = 1 % nslot;
    if (A.use_glds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sync();

    struct Half {                                         // one of the packed problems, for the shared epilogue
        const Lane& L;
        int h;
        __device__ __forceinline__ float level_value(int m, int) const { return pk_half(L.level(m), h); }
    };

    auto one_step = [&]() {
        if (a_u == A.issue_at && k_u + 1 < nx) {
            sync();                                       // every wave has left the slot that is overwritten now
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
            // the accumulators are cleared through L.keep; a pair that overflowed gets the explicit reset (0 * inf is NaN)
            if (!A.keep_reset || !(fabsf(pk_half(L.ktop, 0)) <= 3.0e38f) || !(fabsf(pk_half(L.ktop, 1)) <= 3.0e38f)) L.reset();
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
        L.keep = pk_splat<V>((dummy && A.keep_reset) ? 0.f : 1.f);
        // if lane 0 opens a new x at the next step, its record (requested issue_at steps into this x) must have landed
        if (a_u == 0) {
            if (A.use_glds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sync();                                       // ... the pieces of every wave
        }
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

template <typename V, int W, int G, int C, int D, int M, int MODE>
hipError_t seq_pk2_launch(const SeqGramArgs& A, int ntasks, size_t lds_bytes, hipStream_t stream) {
    if (ntasks <= 0) return hipSuccess;
    auto kern = seq_pk2_kernel<V, W, G, C, D, M, MODE>;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(ntasks), dim3(W * 64), lds_bytes, stream, A);
    return hipGetLastError();
}

}  // namespace gpsig
