// seq_gram_kernel.hpp -- gfx950 kernel for the sequence-vs-sequence signature-kernel Gram
// (SignatureKernel._K_seq / _K_seq_diag + the normalise / weight / level-sum epilogue of
// SignatureKernel.K, gpsig/kernels.py:188-237 and :430-476).
//
// One 64-lane workgroup (= one wavefront) per task.  A task is a block of 64/G y-side sequences held
// in registers (each pair group of G lanes owns one y; lane lam owns C consecutive lattice columns)
// against a run of x-side sequences streamed through a small LDS ring.  Lanes of a group run the
// lattice-row recursion of seq_core.hpp skewed by one step per lane, so the only cross-lane traffic
// is one DPP shift (row_shr:1 / wave_shr:1) of M carries + (M-1) diagonal terms per step, and pairs
// follow each other without draining the pipeline.  Levels are normalised, weighted and summed in
// registers at each pair boundary; one 8-byte store (two with the mirror) leaves the chip per pair.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <limits>
#include <type_traits>

#include "seq_args.hpp"
#include "seq_core.hpp"

namespace gpsig {

template <int G>
__device__ __forceinline__ int dpp_shr1_i32(int v) {
    // lane l receives lane l-1's value; the first lane of each group of G receives 0
    if constexpr (G == 16) return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    else return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);                      // wave_shr:1
}
template <int G>
__device__ __forceinline__ double shr1(double v) {
    int lo = dpp_shr1_i32<G>(__double2loint(v)), hi = dpp_shr1_i32<G>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int G>
__device__ __forceinline__ float shr1(float v) {
    return __int_as_float(dpp_shr1_i32<G>(__float_as_int(v)));
}

// EXACT: num_levels == MMAX is a compile-time constant (no level branches in the step);
// otherwise any num_levels <= MMAX is accepted at run time.
// OMAX == 0: first-order algorithm (signature_algs.py:8-35).  OMAX >= 2: higher-order algorithm
// (signature_algs.py:37-74) for any run-time order <= OMAX.
// KIND >= 0: the base kernel at compile time (built for the RBF kernel with differences, exact shapes): the C + 1 evaluations
// of a step interleave instead of queueing behind a switch -- 5 % at the headline shape, 18 % at one wavefront per SIMD.
#define SEQ_FAST_RBF(T, MODE, OMAX, KIND) (sizeof(T) == 8 && (KIND) == BASE_RBF && (MODE) == MODE_PT_DIFF)      // (OMAX > 0: round 6's exact higher-order instances)
// ... and the Matern families (round 5): prescaled records too, seq_step_matern_prescaled in seq_core.hpp
#define SEQ_FAST_MATERN(T, MODE, OMAX, KIND) (sizeof(T) == 8 && seq_is_matern(KIND) && (MODE) == MODE_PT_DIFF)      // (OMAX > 0: round 6's exact higher-order instances)
// The float64 RBF instances (round 5): two steps per loop trip -- the hand-over words alternate registers instead of being copied back, 12
// v_mov_b64 of the 170 vector instructions of a step's body at the headline shape (tools/isa_loops.py --blocks: 170 -> 157) -- compiled for TWO
// wavefronts per SIMD: the doubled live ranges need 197 registers, and held to the 168 of three wavefronts the body spills (69 ms).  Same box,
// alternating processes, configs[1] with SignatureRBF (profiles/r05_ab_c2rbf.txt): 44.4 ms as built in rounds 2-4 (one step per trip, three
// waves), 45.5 one step / two waves, 42.7 two steps / two waves.  SEQ_RBF_UNROLL2 / SEQ_RBF_WAVES rebuild the other forms for A/B runs.
#ifndef SEQ_RBF_UNROLL2
#define SEQ_RBF_UNROLL2 1
#endif
#ifndef SEQ_RBF_WAVES
#define SEQ_RBF_WAVES 2
#endif
// STASH (round 5, the float64 instances whose pairs the fused reverse kernel takes -- RBF at compile time, and the run-time-kind instances for the
// Matern families): the kernel also writes what that reverse pass needs of
// this recursion -- every lattice row's totals of levels 1 .. M-1 (the last lane's hand-over words) and every lane's Q's when its pair ends
// (SeqGramArgs::stash) -- so that the backward call starts at the turn of the sweeps instead of repeating the forward one.
template <typename T, int G, int C, int D, int MMAX, int MODE, bool EXACT, int OMAX = 0, int KIND = -1, bool STASH = false>
__global__ __launch_bounds__(64, ((SEQ_FAST_RBF(T, MODE, OMAX, KIND) || SEQ_FAST_MATERN(T, MODE, OMAX, KIND)) && OMAX == 0) && C * D <= 32 ? SEQ_RBF_WAVES : ((MODE != MODE_INC && OMAX == 0 && C * D <= 32) ? 2 : 1)) void seq_gram_kernel(const SeqGramArgs A) {
    static_assert(!STASH || (sizeof(T) == 8 && MODE == MODE_PT_DIFF && OMAX == 0 && EXACT && MMAX >= 2),
                  "the stash is written by exact float64 instances of the first-order algorithm on points with differences");
    static_assert(G == 16 || G == 64, "pair group is a DPP row or the whole wave");
    static_assert((D * sizeof(T)) % 16 == 0, "record rows are read with 16-byte LDS loads");
    using Lane = typename std::conditional<OMAX == 0, SeqLane<T, C, D, MMAX, MODE>, SeqLaneHO<T, C, D, MMAX, (OMAX > 0 ? OMAX : 1), MODE>>::type;
    constexpr int VEC = 16 / sizeof(T);                  // elements per 16-byte piece
    typedef T vecT __attribute__((ext_vector_type(VEC)));

    // float64 RBF at compile time: prescaled records + table-driven exp (seq_step_rbf_prescaled in seq_core.hpp); the host
    // prepares the records accordingly whenever it launches such an instance (SeqPlanned::rbf_prescaled in api.hip)
    constexpr bool FAST_RBF = SEQ_FAST_RBF(T, MODE, OMAX, KIND), FAST_MATERN = SEQ_FAST_MATERN(T, MODE, OMAX, KIND);
    static_assert(!FAST_MATERN || SEQ_EXP256, "the Matern instances use the 256-entry table");
    __shared__ double etab[FAST_RBF || FAST_MATERN ? SEQ_ETAB_N : 1];
    if constexpr (FAST_RBF || FAST_MATERN) {
        if constexpr (SEQ_EXP256) exp_tab256_fill(etab, int(threadIdx.x), 64);
        else exp_tab_fill(etab, int(threadIdx.x), 64);
    }

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* const zero_row = reinterpret_cast<T*>(smem_raw);   // RS elements of zeros (rows of idle lanes); LaneCtl offsets start here
    T* const ring = zero_row + A.RS;                       // nslot slots of slot_elems

    const int lane = threadIdx.x;
    const int lam = lane & (G - 1);
    const int grp = lane / G;
    const SeqTask tk = A.tasks[blockIdx.x];
    int64_t stash_pair0 = 0;
    if constexpr (STASH) {
        const SeqTask pz = A.stash_pair0[blockIdx.x];
        stash_pair0 = (int64_t(pz.x0) << 32) | int64_t(uint32_t(pz.y0));
    }
    const int M = EXACT ? MMAX : A.M;
    const int R1 = A.R1, RS = A.RS, nslot = A.nslot, nx = tk.nx;
    const T* const xrec = static_cast<const T*>(A.xrec);
    const T* const yrec = static_cast<const T*>(A.yrec);

    if (lane < RS) zero_row[lane] = T(0);

    // ---- y side: this lane's C record rows of sequence j -------------------------------------
    const int64_t j = int64_t(tk.y0) + grp;
    const bool jvalid = j < A.N2;
    Lane L;
    L.init();
#pragma unroll
    for (int r = 0; r < C; ++r) {
        const int row = C * lam + r;
        const bool ok = jvalid && row < A.R2;
        const T* src = yrec + (ok ? j * A.yrec_stride + int64_t(row) * RS : 0);
        T ys = T(0);
#pragma unroll
        for (int f = 0; f < D; f += VEC) {
            vecT v = ok ? *reinterpret_cast<const vecT*>(src + f) : vecT(T(0));
#pragma unroll
            for (int e = 0; e < VEC; ++e) { L.y[r][f + e] = v[e]; ys = fma(v[e], v[e], ys); }
        }
        L.y2[r] = FAST_RBF ? T(-0.5) * ys : ys;
    }
    // real lattice columns among the owned ones (point modes; see seq_core.hpp)
    const int rlo = (lam == 0) ? 1 : 0;
    int rhi = A.R2 - C * lam;
    rhi = rhi < 0 ? 0 : (rhi > C ? C : rhi);

    // ---- x side staging ------------------------------------------------------------------------
    auto stage = [&](int p, int slot) {   // copy the record of x number p of the run into ring slot `slot`
        int64_t i = int64_t(tk.x0) + p;
        if (i >= A.N1) i -= A.N1;                         // circulant runs wrap around
        const T* src = xrec + i * A.xrec_stride;
        T* dst = ring + int64_t(slot) * A.slot_elems;
        const int pieces = A.slot_elems / VEC;            // multiple of 64
        if (A.use_glds) {
            for (int c = 0; c < pieces; c += 64)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + int64_t(c + lane) * VEC),
                    (__attribute__((address_space(3))) void*)(dst + int64_t(c) * VEC), 16, 0, 0);
        } else {
            for (int c = lane; c < pieces; c += 64)
                reinterpret_cast<vecT*>(dst)[c] = reinterpret_cast<const vecT*>(src)[c];
        }
    };
    stage(0, 0);
    if (A.pred == PRED_DIAG_OWN)                          // one record per pair group, side by side
        for (int g = 1; g < 64 / G; ++g) stage(g, g);

    LaneCtl ctl;
    ctl.init(lam, RS);
    if (A.pred == PRED_DIAG_OWN) ctl.sbase += grp * A.slot_elems;
    const int ring_elems = nslot * A.slot_elems;
    const T p0 = T(A.p0), p1 = T(A.p1);

    // left-neighbour reads: one DPP shift per 32-bit half, issued where the value is consumed
    struct DevNbr {
        const Lane& L;
        __device__ __forceinline__ T cin(int m) const { return shr1<G>(L.s[m]); }
        __device__ __forceinline__ T kleft() const { return shr1<G>(L.klast); }
        __device__ __forceinline__ T win(int m, int r) const {
            if constexpr (Lane::HIGHER_ORDER) return shr1<G>(L.w[m][r]); else return T(0);
        }
    };

    // Lane 0 of each group starts x number k at step k*R1: its record must be resident by then.  The next
    // record is requested issue_at steps later, into a slot the slowest lane no longer reads (seq_ring in
    // seq_args.hpp).  After the last x, G more steps let lane lam emit its last pair at step nx*R1 + lam.
    const int nsteps = nx * R1 + G;
    int a_u = 0, k_u = 0, slot_next = 1 % nslot;          // wave-uniform position of lane 0
    // x-side record row of a lane for the step described by `cc` (16-byte LDS reads; rows are RS = D + pad apart,
    // so the 16 lanes of a pair group hit 16 different bank groups)
    auto load_row = [&](const LaneCtl& cc, T (&dst)[D]) {
        const T* rowp = zero_row + cc.rowoff;
#pragma unroll
        for (int f = 0; f < D; f += VEC) {
            vecT v = *reinterpret_cast<const vecT*>(rowp + f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) dst[f + e] = v[e];
        }
    };
    // (Requesting the next step's row one step ahead was tried: 2*D more live registers cost the flagship shape a wave
    // per SIMD and 7 % of its speed -- profiles/r01_ab_variants.txt.)
    if (A.use_glds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    auto one_step = [&]() {
        if (a_u == A.issue_at && k_u + 1 < nx) {
            stage(k_u + 1, slot_next);
            if (++slot_next == nslot) slot_next = 0;
        }
        if (++a_u == R1) { a_u = 0; ++k_u; }

        // pair boundary: the pair that just finished is complete in the last lane of the group
        if (ctl.begin_step(nx, R1, RS, A.slot_elems, ring_elems)) {
            if (lam == G - 1 && ctl.p >= 1 && jvalid) {
                int64_t i = int64_t(tk.x0) + (ctl.p - 1);
                if (i >= A.N1) i -= A.N1;
                T* const out = static_cast<T*>(A.out);
                seq_emit<T>(L, A, i, j, M, [&](int64_t off, T v) { out[off] = v; });
            }
            if constexpr (STASH) {
                if (ctl.p >= 1 && jvalid) {              // this lane's Q's of the pair it has just finished
                    double* qd = A.stash + ((stash_pair0 + (ctl.p - 1)) * (64 / G) + grp) * A.stash_stride
                                 + int64_t(A.R1 - 1) * (MMAX - 1) + lam * ((MMAX - 1) * C);
#pragma unroll
                    for (int m = 0; m < MMAX - 1; ++m)
#pragma unroll
                        for (int r = 0; r < C; ++r) __builtin_nontemporal_store(L.q[m][r], qd + m * C + r);
                }
            }
            // first-order lanes clear their accumulators through L.keep (below); only a pair that overflowed needs the explicit
            // reset, so that its inf / NaN does not outlive it in this lane
            if constexpr (Lane::HIGHER_ORDER) L.reset();
            else if (!A.keep_reset || !(fabs(L.ktop) <= std::numeric_limits<T>::max())) L.reset();
        }

        T xr[D];
        load_row(ctl, xr);
        T hx = T(0);
        if constexpr (FAST_RBF) hx = zero_row[ctl.rowoff + D];            // -|x'|^2 / 2, the record row's spare column
        const bool dummy = ctl.row0;
        if constexpr (!Lane::HIGHER_ORDER) L.keep = (dummy && A.keep_reset) ? T(0) : T(1);     // row 0 of an x (or an idle lane): see SeqLane::keep

        // if lane 0 opens a new x at the next step, its record (requested issue_at steps into this x) must have landed
        if (a_u == 0 && A.use_glds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        if constexpr (FAST_RBF && OMAX > 0) seq_step_rbf_prescaled_ho(L, DevNbr{L}, xr, hx, etab, M, EXACT ? OMAX : A.order, dummy, rlo, rhi);
        else if constexpr (FAST_RBF) seq_step_rbf_prescaled(L, DevNbr{L}, xr, hx, etab, M, dummy, rlo, rhi);
        else if constexpr (FAST_MATERN && OMAX > 0) seq_step_matern_prescaled_ho<KIND>(L, DevNbr{L}, xr, etab, M, EXACT ? OMAX : A.order, dummy, rlo, rhi);
        else if constexpr (FAST_MATERN) seq_step_matern_prescaled<KIND>(L, DevNbr{L}, xr, etab, M, dummy, rlo, rhi);
        else if constexpr (KIND == BASE_SPECTRAL && OMAX == 0 && MODE != MODE_INC) seq_step_spectral(L, DevNbr{L}, xr, A.spec, int(A.p0), int(A.p1), M, dummy, rlo, rhi);
        else seq_step(L, DevNbr{L}, xr, M, A.order, dummy, rlo, rhi, KIND >= 0 ? KIND : A.kind, p0, p1);
        if constexpr (STASH) {
            if (lam == G - 1 && !dummy && jvalid && ctl.p >= 0 && ctl.p < nx) {      // the row totals of the lattice row just swept
                double* rt = A.stash + ((stash_pair0 + ctl.p) * (64 / G) + grp) * A.stash_stride
                             + int64_t(A.R1 - ctl.left - 1) * (MMAX - 1);
#pragma unroll
                for (int m = 0; m < MMAX - 1; ++m) __builtin_nontemporal_store(L.s[m], rt + m);      // written once, read by another launch
            }
        }
        ctl.end_step();
    };
    // two steps per trip: the loop-carried hand-over words (s, qold) alternate registers instead of being copied
    // back every step.  An odd step count is rounded up; the extra step finds every lane past its last pair.
    // (only where it pays: the point-kernel and higher-order bodies are large enough to lose a wave per SIMD to
    // the doubled live ranges)
    if constexpr ((MODE == MODE_INC && OMAX == 0) || (SEQ_RBF_UNROLL2 && (FAST_RBF || FAST_MATERN) && OMAX == 0 && C * D <= 32)) {
        for (int t = 0; t < nsteps; t += 2) {
            one_step();
            one_step();
        }
    } else {
        for (int t = 0; t < nsteps; ++t) one_step();
    }
}

template <typename T, int G, int C, int D, int MMAX, int MODE, bool EXACT, int OMAX = 0, int KIND = -1, bool STASH = false>
hipError_t seq_gram_launch(const SeqGramArgs& A, int ntasks, size_t lds_bytes, hipStream_t stream) {
    if (ntasks <= 0) return hipSuccess;
    auto kern = seq_gram_kernel<T, G, C, D, MMAX, MODE, EXACT, OMAX, KIND, STASH>;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(ntasks), dim3(64), lds_bytes, stream, A);
    return hipGetLastError();
}

}  // namespace gpsig
