// seq_args.hpp -- launch arguments of the seq-gram kernel, task list construction and the pair
// epilogue.  HIP-free: shared by the gfx950 kernel, the host API and the CPU lock-step emulator
// used by the test-suite.
#pragma once

#include <stdint.h>
#include <vector>

#include "seq_core.hpp"

namespace gpsig {

struct SeqTask {
    int32_t y0;   // first y-side sequence of the block
    int32_t x0;   // first x-side sequence of the run
    int32_t nx;   // run length
};

// PRED_DIAG_OWN: the diagonal with several pair groups per wavefront, every group sweeping ITS OWN sequence (task.nx == 1, the
// records of x0 .. x0 + 64/G - 1 staged side by side in the ring) instead of all groups sweeping the same 64/G sequences with one
// emitted pair each (PRED_DIAG) -- a quarter of the work at G = 16.  A value of `pred` rather than a field of its own: the pair
// kernels read `pred` at pair boundaries anyway, and one more live scalar cost the headline instance its third wavefront per
// SIMD (168 -> 171 VGPRs, 24.7 -> 26.9 ms).
enum : int { PRED_ALL = 0, PRED_CIRCULANT = 1, PRED_DIAG = 2, PRED_DIAG_OWN = 3 };

struct SeqGramArgs {
    const void* xrec;       // x-side records: N1 x rec_stride elements
    const void* yrec;       // y-side records: N2 x rec_stride_y elements
    const SeqTask* tasks;
    int64_t N1, N2;
    int64_t xrec_stride, yrec_stride;  // elements between consecutive sequences' records
    int32_t R1, R2;         // record rows per sequence on each side
    int32_t RS;             // elements between consecutive record rows (D + pad)
    int32_t M;
    int32_t order;          // 1, or the reference's `order` for the higher-order kernels
    int32_t nslot;          // LDS ring depth
    int32_t issue_at;       // step within an x at which the next x's record is requested (seq_ring)
    int32_t slot_elems;     // elements per ring slot (>= R1*RS, multiple of 128 so a slot is whole 1 KiB DMA pieces for fp64)
    int32_t kind;           // base kernel (point modes)
    double p0, p1;
    // epilogue:  v_m = (K_m + [i==j] * jitter_diag) * ax[m][i] * by[m][j],  m = 0..M  (K_0 = 1)
    void* out;
    int64_t si, sj, sm;     // element strides of the x index, y index, level index in `out`
    const void* ax;         // (N1, M+1) sequence-major, or NULL (== 1)
    const void* by;         // (N2, M+1) sequence-major, or NULL (== 1)
    double jitter_diag;
    int32_t sum_levels;     // 1: out[i*si + j*sj] = sum_m v_m ; 0: out[m*sm + i*si + j*sj] = v_m
    int32_t pred;           // PRED_*
    int32_t mirror;         // also store at (j, i)
    int32_t use_glds;       // stage x records with global_load_lds (LDS DMA) instead of load + ds_write
    int32_t compact;        // PRED_CIRCULANT only: owned entries of row j packed as out[j*sj + (N/2 - (j-i) mod N)], i.e. row j's
                            // N/2+1 owned columns j-N/2 .. j side by side (multi-GPU row blocks: half the bytes to gather)
    int32_t keep_reset;     // 1: first-order lanes clear their accumulators through SeqLane::keep, 0: explicit reset() at pair boundaries
    const double* spec;     // BASE_SPECTRAL: alpha[Q], omega[Q][D], gamma[Q][D] (Q = p0, family = p1, D = the kernel's padded width)
    // STASH instances only (round 5): what the reverse pass needs of the forward recursion, kept instead of recomputed -- per pair the row
    // totals of levels 1 .. M-1 of every lattice row, then every lane's final Q's; pair k of task t, group g, sits at
    // stash + ((pair0(t) + k) * (64 / G) + g) * stash_stride  (doubles; layout in grad_fused_kernel.hpp: fused_stash_stride), pair0(t) packed
    // into stash_pair0[t] as (x0 << 32 | y0) -- the task lists' device cache holds SeqTask records
    double* stash;
    const SeqTask* stash_pair0;
    int64_t stash_stride;
};


// Pair epilogue (gpsig/kernels.py:430-433 / :463-469 normalisation, :471 sigma*variances, :473-476
// level sum), applied to the finished pair held by lane state L.  `store(offset, value)` writes one
// element of `out`.
template <typename T, class Lane, class Store>
GPSIG_HD void seq_emit(const Lane& L, const SeqGramArgs& A, int64_t i, int64_t j, int M, Store store) {
    bool emit = true;
    int64_t cdlt = 0;
    if (A.pred == PRED_CIRCULANT) {
        const int64_t N = A.N1, H = N / 2;
        int64_t dlt = j - i;
        if (dlt < 0) dlt += N;
        emit = dlt < H || (dlt == H && ((N & 1) || i < j));
        cdlt = H - dlt;
    } else if (A.pred == PRED_DIAG) {
        emit = (i == j);
    } else if (A.pred == PRED_DIAG_OWN) {
        i = j;                                   // the group swept its own sequence (the caller's run index is meaningless here)
    }
    if (!emit) return;
    const T* ax = A.ax ? static_cast<const T*>(A.ax) + i * (M + 1) : nullptr;
    const T* by = A.by ? static_cast<const T*>(A.by) + j * (M + 1) : nullptr;
    const T dj = (i == j) ? T(A.jitter_diag) : T(0);
    int64_t o1 = i * A.si + j * A.sj, o2 = j * A.si + i * A.sj;
    if (A.compact) o1 = j * A.sj + cdlt;
    const bool mir = A.mirror && i != j;
    T acc = T(0);
    for (int m = 0; m <= M; ++m) {
        T v = (m == 0 ? T(1) : L.level_value(m, M)) + dj;
        if (ax) v *= ax[m];
        if (by) v *= by[m];
        if (A.sum_levels) {
            acc += v;
        } else {
            store(o1, v);
            if (mir) store(o2, v);
            o1 += A.sm;
            o2 += A.sm;
        }
    }
    if (A.sum_levels) {
        store(o1, acc);
        if (mir) store(o2, acc);
    }
}

// ---- host-side planning ----------------------------------------------------------------------
// LDS ring of x-side records.  Lane lam of a pair group starts x number k at step k*R1 + lam, so while lane 0
// is already on x_k the slowest lane (G-1) still reads x_{k-1} for G-1 more steps.  The record of x_{k+1} is
// requested at step k*R1 + issue_at.  With issue_at = G the slot of x_{k-1} is free by then, so two slots
// suffice as long as the copy has R1 - G steps to land (R1 >= 2G); shorter records are requested at the
// start of x_k instead and the ring is deepened until (nslot - 2) * R1 >= G - 1.
struct SeqRing {
    int nslot, issue_at;
};
inline SeqRing seq_ring(int G, int R1) {
    if (R1 >= 2 * G) return SeqRing{2, G};
    return SeqRing{2 + (G - 1 + R1 - 1) / R1, 0};
}
inline int seq_ring_depth(int G, int R1) { return seq_ring(G, R1).nslot; }

// Task list.  pred == PRED_ALL: every (x, y) pair of an N1 x N2 cross Gram.  PRED_CIRCULANT (N1 == N2,
// same sequences): each unordered pair once -- y block [y0, y0+ypb) against the x window
// [y0 - N/2, y0 + ypb) taken modulo N, so every block has the same amount of work.  PRED_DIAG: the
// block-diagonal only.  Runs are cut into pieces of at most `max_run` x's; shard (index, count)
// keeps every count-th task (tasks are independent).
// [y_begin, y_end): restrict to the y blocks of that range (y_begin a multiple of ypb); (0, -1) = all.
inline std::vector<SeqTask> seq_build_tasks(int64_t N1, int64_t N2, int ypb, int pred, int max_run,
                                            int shard_index, int shard_count, int64_t y_begin = 0, int64_t y_end = -1) {
    std::vector<SeqTask> t;
    int64_t counter = 0;
    auto push = [&](int64_t y0, int64_t x0, int64_t nx) {
        for (int64_t o = 0; o < nx; o += max_run) {
            int64_t n = nx - o < max_run ? nx - o : max_run;
            int64_t xs = x0 + o;
            if (pred == PRED_CIRCULANT) xs %= N1;
            if ((counter++ % shard_count) == shard_index) t.push_back(SeqTask{int32_t(y0), int32_t(xs), int32_t(n)});
        }
    };
    if (y_end < 0 || y_end > N2) y_end = N2;
    for (int64_t y0 = y_begin; y0 < y_end; y0 += ypb) {
        if (pred == PRED_ALL) {
            push(y0, 0, N1);
        } else if (pred == PRED_DIAG) {
            int64_t hi = y0 + ypb < N1 ? y0 + ypb : N1;
            if (y0 < hi) push(y0, y0, hi - y0);
        } else {
            const int64_t N = N1, H = N / 2;
            int64_t nx = H + ypb;
            if (nx > N) nx = N;
            int64_t x0 = ((y0 - H) % N + N) % N;
            push(y0, x0, nx);
        }
    }
    return t;
}

}  // namespace gpsig
