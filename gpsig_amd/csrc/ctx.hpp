// ctx.hpp -- the gpsig_ctx object and the host-side helpers shared by the translation units that implement
// the C ABI (api.hip: evaluation, grad_api.hip: gradients).
#pragma once
#include <atomic>

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gpsig_hip.h"
#include "seq_args.hpp"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

enum BufId {
    B_IN0, B_IN1, B_IN2,          // host-mode input staging
    B_OUT0, B_OUT1, B_OUT2,       // host-mode output staging
    B_REC0, B_REC1,               // seq-gram records
    B_DLEV0, B_DLEV1,             // diagonal levels
    B_FAC0, B_FAC1,               // per-sequence factors
    B_TASKS, B_W, B_XT, B_ZT, B_ZS, B_ZL, B_ZN, B_TMP0, B_TMP1,
    B_LS, B_LR0, B_LR1, B_LR2, B_LR3, B_LR4, B_LR5, B_LR6, B_LR7, B_LR8,
    B_GR0, B_GR1, B_GR2, B_GR3, B_GR4, B_GR5, B_GR6, B_GR7,   // gradient path scratch
    B_GR7B, B_TW0, B_TW1, B_TW2,
    B_SF0, B_SF1, B_SF2, B_SF3, B_SF4, B_SF5,                               // explicit level features (sig_feat_kernel.hpp): both sides, partial products, level diagonals                              // weighted tensor-vs-sequence sums: partial factor gradients; level arrays of the fallback
    B_SPEC,                                                    // spectral base-kernel table
    B_TQ,                                                      // item counters of the Kzx tile kernel's persistent launch
    B_STASH,                                                   // what the fused reverse kernel needs of the forward recursion (gpsig_seq_gram_levels_stash)
    B_WD0, B_WD1, B_WD2, B_WD3, B_WD4, B_WD5, B_WD6, B_WD7, B_WD8, B_WD9, B_WD10,   // wide state spaces (wide_api.hip): augmented rows, kernel-argument chunks, their adjoints, lattice states
    B_COUNT
};

inline thread_local std::string g_create_error;

// A task list depends only on a handful of integers; the device copy is reused while they do not change (no host rebuild, no
// upload, no host-device synchronisation per call).
struct TaskCache {
    int64_t key[10] = {0};
    int ntasks = 0;
    bool valid = false;
    bool match(const int64_t (&k)[10]) const {
        if (!valid) return false;
        for (int i = 0; i < 10; ++i) if (key[i] != k[i]) return false;
        return true;
    }
    void set(const int64_t (&k)[10], int n) { for (int i = 0; i < 10; ++i) key[i] = k[i]; ntasks = n; valid = true; }
};

struct TaskSlot {
    TaskCache tc;
    DevBuf buf;
    uint64_t stamp = 0;
    int64_t aux = 0;             // what the builder returned (the number of pairs the list covers)
};
constexpr size_t TASK_SLOTS = 16;

// Stash generations are drawn from ONE counter per process (round 6): a descriptor handed to another context -- a backward pass on a different
// stream than its forward pass, a C caller mixing contexts -- can then never match that context's own generation by coincidence.
inline int64_t next_stash_generation() {
    static std::atomic<int64_t> counter{0};
    return ++counter;
}

struct gpsig_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    int ptr_mode = GPSIG_PTR_HOST;
    int shard_i = 0, shard_n = 1;
    int use_glds = 1;
    int allow_exact = 1;
    int keep_reset = 1;           // pair kernels: accumulators cleared through SeqLane::keep (1) or by reset() at pair boundaries (0)
    int f32_pack = 2, f32_waves = 0;   // seq_pk2_kernel variant: y sequences per pair group (1 / 2); wavefronts per workgroup (1 / 4, 0 = by launch size)
    int allow_pk2 = 1;            // float32: the packed two-sequence kernels (seq_pk2_kernel.hpp) where they are built
    int max_run = 0;
    int tens_lanes = -1;   // -1 auto, 0 sequence lanes, 1 tensor lanes
    int grad_scratch_mb = 4096;   // lattice scratch of one gradient launch
    int matern_fast = 1;          // float64 sequence Grams of the Matern families: 1 = compile-time instances on prescaled records where built, 0 = run-time kind
    int grad_fused_piece = 0;     // streamed sequences per workgroup of the backward sweep from the stash (0: 16)
    int grad_stash_mb = 4096;     // gpsig_seq_gram_levels_stash keeps at most this much for the backward call (0: never)
    bool stash_want = false;      // set around the forward launch by gpsig_seq_gram_levels_stash; launch_seq fills stash_desc if it wrote one
    int64_t stash_desc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // generation (0: none), pred, max_run, ypb, tasks, pair slots, stride, lattice rows
    int64_t stash_gen = 0;        // the generation of THIS context's stash: drawn from next_stash_generation(), unique in the process
    int grad_impl = 0;            // 0: planner's choice, 1: one pair per thread + stored lattice, 2: one pair per thread scratch-free (tensor-vs-seq),
                                  // 3: wavefront kernel + stored lattice, 4: scratch-free wavefront kernel wherever it is built
    void* blas_handle = nullptr;  // rocBLAS handle of gpsig_lr_whitening (lowrank_solver.hip), created at first use
    int tvs_tile = -1;            // Kzx tile kernel (tvs_tile_kernel.hpp): -1 where it is built, 0 never, 1 also below 32 tensors
    int wide = -1;                // wide state spaces (wide_api.hip: kernel arguments by dgemm, fused map / difference / recursion kernels): -1 where the exact-shape
                                  // kernels are not built (more than 8 columns for Kzx, more than 32 for the sequence lattices), 0 never, 1 wherever built
    int wide_contract = 1, wide_lat_waves = -1, wide_sym_fold = 1, ho_g32 = -1, wide_few_cols = 0, tvs_grad_matern = 1, wide_o1_sweeps = 1;        // the reverse pass's two contractions for narrow rows: 1 = one hand-written pass over the adjoint array (wide_contract_kernel), 0 = rocBLAS dgemms
    int wide_chunk_mb = 0;        // its argument chunk in HBM (0: a quarter of the gradient scratch budget)
    int tvs_features = -1;        // Kzx of the linear / cosine kernel as one product of level features: -1 where a time model prefers it, 0 never, 1 wherever built
    int tvs_tile_nw = 0;          // its waves per workgroup: 0 = planner's choice
    int tens_tile = 1;            // Kzz in 16 x 16 tiles with the tensors staged in LDS (tens_gram_tile_kernel); 0: one gathering thread per entry
    int spectral_wave = 1;        // SignatureSpectral sequence kernels: 1 = wavefront kernels where built, 0 = one pair per thread (round 1)
    int diag_own = 1;             // diagonal pass: every pair group sweeps its own sequence (SeqGramArgs::diag_own); 0: round-1 form, for A/B runs
    int lr_gemm = 1;              // low-rank Gram products: 1 = 128 x 128 tiles staged through LDS, 0 = fragments straight from L2 (round 1)
    int lr_fused_pad = 1;         // row stride of its LDS arrays beyond the time steps rounded up to 64, in doubles (A/B runs)
    int lr_fused_variant = 0;     // its workgroup size / unrolling (lr_fused_inst.hip), for A/B runs
    int lr_fused = 1;             // low-rank sequence features: 1 = the fused kernel where a sequence's arrays fit LDS, 0 = one kernel per op
    double* tvs_aux_out = nullptr;   // set by gpsig_tens_vs_seq_weighted around its launch: where the tile kernel leaves the chain totals
    bool tvs_aux_written = false;    // ... and whether it did (only the tile kernel does)
    int tvs_grad_tile = 1;        // tensor-vs-sequence reverse pass: 1 = the tile kernel (tvs_grad_tile_kernel.hpp) where built, 0 = the round-1 kernels
    int sig_features = -1;        // SignatureLinear Grams as a contraction of explicit level features: -1 where cheaper, 0 never, 1 wherever built
    int sf_keep = 0;                     // keep the feature matrix between calls ("sig_features_keep")
    bool sf_valid = false;
    const void* sf_X = nullptr; const void* sf_phi = nullptr; uint64_t sf_key = 0;
    int sig_features_grad = -1;          // gradients of SignatureLinear's sequence levels through the feature contraction (sig_feat_grad_api.hip): -1 where cheaper, 0 never, 1 wherever built
    int lr_grad_threads = 1024;          // workgroup size of the low-rank reverse kernel (lr_grad_kernel.hpp): 1024 or 512
    int sig_graded = 1;                  // the contraction's last depth piece cut into finer ones (sig_piece_bounds); 0: equal pieces (round 3)
    int sig_gemm_dma = 1;                // the contraction's slabs by LDS-DMA with fragments prefetched across the barrier (0: register-staged form)
    int lr_jacobi = 1;            // gpsig_lr_draw: eigendecomposition of the landmark Gram by the one-workgroup Jacobi kernel (c <= 64), 0: rocSOLVER
    int tvs_zreg = -1;            // tensor-lane gradient: components in registers (1) or LDS (0); -1 = planner's choice
    std::string err;
    DevBuf buf[B_COUNT];
    std::vector<gpsig::SeqTask> host_tasks;
    std::vector<TaskSlot> task_slots;      // device-resident task lists, least recently used one replaced (task_list())
    uint64_t task_clock = 0;
    std::vector<double> last_weights;      // what B_W currently holds
    std::vector<double> last_ls;           // what B_LS currently holds (lengthscales of a state space wider than MAX_FEATURES)
    void* ls_base = nullptr;
    std::vector<double> last_spec;         // what B_SPEC currently holds (SignatureSpectral's parameter table)
    void* spec_base = nullptr;
    // low-rank mode: what B_LR0 / B_LR1 currently hold (keyed by content: the random objects of an evaluation are handed to several
    // calls -- tensor features, sequence features, products -- and every upload was a host synchronisation)
    uint64_t lr_hash = 0;                  // 0: nothing uploaded
    void* lr_base = nullptr;               // the B_LR0 block the offsets below refer to
    std::vector<size_t> lr_offsets;        // byte offsets of the uploaded arrays inside it, in lr_upload's order
    int64_t lr_off_key[3] = {-1, -1, -1};  // (M, c, r) of the level offsets in B_LR1
    void* lr_off_base = nullptr;
    // the projections of the training path's low-rank entry points (lr_grad_api.hip), kept in B_LR8 by content
    uint64_t lrg_hash = 0;
    void* lrg_base = nullptr;
    std::vector<size_t> lrg_offsets;
    // timing of the pair-recursion launches
    std::vector<hipEvent_t> ev;     // pairs (start, stop)
    size_t ev_used = 0;
    int64_t t_launches = 0, t_pairs = 0;
    const char* t_kernel = nullptr;      // which kernel the timed launches were, where it is not the pair recursion (gpsig_timing_info)
    double t_flops = 0.0;                // multiply-adds x 2 those launches executed on the matrix cores
    // HIP-graph capture (gpsig_graph_begin .. gpsig_graph_end): launches only -- nothing may allocate, upload or synchronise
    bool capturing = false, capture_failed = false;
    uint64_t alloc_gen = 0;         // bumped whenever a scratch buffer moves; a graph replays only against the generation it saw
    // effective shader clock (gpsig_clock_probe_*): one sleeping wavefront on a stream of its own samples s_memtime / s_memrealtime
    hipStream_t probe_stream = nullptr;
    unsigned long long* probe_buf = nullptr;   // device, 2 * probe_cap counters
    int probe_cap = 0, probe_n = 0;
    static constexpr int PROBE_WAVES = 8;      // one sleeping wavefront per XCD (consecutive workgroups go to consecutive XCDs)
    double probe_ghz[PROBE_WAVES] = {};        // the last read: each wave's mean clock and the XCD it sat on
    int probe_xcc[PROBE_WAVES] = {};
    int probe_waves = 0;
    // host-pointer mode: pinned bounce buffers for large transfers (pageable hipMemcpy runs at ~10 GB/s; pinned chunks + a threaded
    // host copy reach several times that), created at first use
    void* pin[2] = {nullptr, nullptr};
    size_t pin_bytes = 0;
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    int pinned_staging = 1;                    // 0: plain hipMemcpyAsync on the caller's pageable memory (A/B runs)
    // gpsig_lr_draw: the projections are drawn on a stream of their own while the landmark Gram is being decomposed
    hipStream_t side_stream = nullptr;
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    volatile int* probe_stop = nullptr;        // pinned host memory the wave polls: raised by gpsig_clock_probe_read
    // the low-rank states drawn on this context (gpsig_lr_draw) that are still alive: gpsig_ctx_destroy detaches them, so that a
    // state destroyed after its context only frees its block instead of touching a stream that no longer exists
    std::vector<struct gpsig_lr_state*> lr_states;
};

struct gpsig_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t alloc_gen = 0;
    gpsig_ctx* ctx = nullptr;
};

inline int fail(gpsig_ctx* c, int code, const char* fmt, ...) {
    char tmp[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof(tmp), fmt, ap);
    va_end(ap);
    if (c) c->err = tmp; else g_create_error = tmp;
    return code;
}

#define HIPCHK(c, expr)                                                                                         \
    do {                                                                                                        \
        hipError_t e__ = (expr);                                                                                \
        if (e__ != hipSuccess) return fail((c), GPSIG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)
#define CHK(expr)                    \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != GPSIG_OK) return rc__; \
    } while (0)

// what a call needs the host for; inside a graph capture that is an error (the call was not run with these shapes and
// hyper-parameters before the capture began)
inline int no_capture(gpsig_ctx* c, const char* what) {
    if (!c->capturing) return GPSIG_OK;
    c->capture_failed = true;
    return fail(c, GPSIG_ERR_INVALID, "graph capture: %s -- run the same calls once before gpsig_graph_begin", what);
}

// Zero fill on the ctx stream.  Inside a graph capture a kernel does it: memset nodes were seen to run out of order with the
// kernel nodes around them when a recorded graph is replayed (ROCm 7.2), which left zeroed records behind a finished prep kernel.
#ifdef __HIPCC__
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void zero_fill_kernel(unsigned long long* p, size_t n8, unsigned char* tail, int ntail) {
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n8; i += stride) p[i] = 0ull;
    if (blockIdx.x == 0 && int(threadIdx.x) < ntail) tail[threadIdx.x] = 0;
}
#else
__global__ void zero_fill_kernel(unsigned long long* p, size_t n8, unsigned char* tail, int ntail);
#endif
#endif
inline int zero_async(gpsig_ctx* c, void* p, size_t bytes) {
    if (!bytes) return GPSIG_OK;
#ifdef __HIPCC__
    if (c->capturing && (reinterpret_cast<uintptr_t>(p) & 7) == 0) {
        const size_t n8 = bytes / 8;
        size_t g = (n8 + 255) / 256;
        if (g < 1) g = 1;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(zero_fill_kernel, dim3(unsigned(g)), dim3(256), 0, c->stream, static_cast<unsigned long long*>(p), n8,
                           static_cast<unsigned char*>(p) + n8 * 8, int(bytes - n8 * 8));
        HIPCHK(c, hipGetLastError());
        return GPSIG_OK;
    }
#endif
    HIPCHK(c, hipMemsetAsync(p, 0, bytes, c->stream));
    return GPSIG_OK;
}

inline int host_sync(gpsig_ctx* c) {
    CHK(no_capture(c, "the call has to wait for the stream"));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GPSIG_OK;
}

inline int ensure(gpsig_ctx* c, int id, size_t bytes, void** out) {
    DevBuf& b = c->buf[id];
    if (bytes > b.cap) {
        CHK(no_capture(c, "a scratch buffer has to grow"));
        ++c->alloc_gen;
        if (b.p) {
            CHK(host_sync(c));   // nothing in flight may still use the old block
            HIPCHK(c, hipFree(b.p));
            b.p = nullptr;
            b.cap = 0;
        }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess) return fail(c, GPSIG_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        b.cap = want;
    }
    *out = b.p;
    return GPSIG_OK;
}

// Device copy of the task list identified by `key` (its defining integers, key[9] = which builder).  On a miss `build` fills
// c->host_tasks (returning a number kept with the list) and the list is uploaded into the least recently used slot.  *tasks is null for an empty list.
template <typename Build>
inline int task_list(gpsig_ctx* c, const int64_t (&key)[10], Build build, const gpsig::SeqTask** tasks, int* ntasks, int64_t* aux = nullptr) {
    TaskSlot* slot = nullptr;
    for (TaskSlot& s : c->task_slots)
        if (s.tc.match(key)) { slot = &s; break; }
    if (!slot) {
        CHK(no_capture(c, "a task list has to be built and uploaded"));
        if (c->task_slots.size() < TASK_SLOTS) {
            if (c->task_slots.capacity() < TASK_SLOTS) c->task_slots.reserve(TASK_SLOTS);
            c->task_slots.emplace_back();
            slot = &c->task_slots.back();
        } else {
            slot = &c->task_slots[0];
            for (TaskSlot& s : c->task_slots)
                if (s.stamp < slot->stamp) slot = &s;
        }
        slot->tc.valid = false;
        ++c->alloc_gen;                          // a recorded graph may still point at what this slot held
        slot->aux = build(c->host_tasks);
        const size_t n = c->host_tasks.size(), bytes = sizeof(gpsig::SeqTask) * n + 64;
        if (bytes > slot->buf.cap) {
            if (slot->buf.p) {
                CHK(host_sync(c));
                HIPCHK(c, hipFree(slot->buf.p));
                slot->buf.p = nullptr;
                slot->buf.cap = 0;
            }
            const size_t want = bytes + bytes / 8 + 256;
            hipError_t e = hipMalloc(&slot->buf.p, want);
            if (e != hipSuccess) return fail(c, GPSIG_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
            slot->buf.cap = want;
        }
        if (n) {
            HIPCHK(c, hipMemcpyAsync(slot->buf.p, c->host_tasks.data(), sizeof(gpsig::SeqTask) * n, hipMemcpyHostToDevice, c->stream));
            CHK(host_sync(c));                   // host_tasks is pageable and reused by the next call
        }
        slot->tc.set(key, int(n));
    }
    slot->stamp = ++c->task_clock;
    *ntasks = slot->tc.ntasks;
    if (aux) *aux = slot->aux;
    *tasks = *ntasks ? static_cast<const gpsig::SeqTask*>(slot->buf.p) : nullptr;
    return GPSIG_OK;
}

// ---- large host <-> device transfers through pinned bounce buffers ------------------------------------------------------
// A numpy array is pageable memory: hipMemcpy stages it through the runtime's own small pinned buffers at about 10 GB/s, which made
// 15 ms of a 42 ms host-pointer evaluation of BASELINE configs[1] (16.8 MB in, 134 MB out).  Here the transfer runs in chunks through
// two pinned buffers of the context: the DMA of chunk k+1 overlaps with a threaded host copy of chunk k.
constexpr size_t PIN_CHUNK_MAX = size_t(64) << 20;
constexpr size_t PIN_MIN = size_t(2) << 20;          // smaller transfers take the plain path
constexpr int PIN_THREADS_MAX = 16;
inline size_t pin_chunk() {                           // GPSIG_PIN_CHUNK_MB / GPSIG_PIN_THREADS: for A/B runs
    static const size_t v = [] { const char* e = getenv("GPSIG_PIN_CHUNK_MB"); size_t m = e ? size_t(atoi(e)) : 16; if (m < 1) m = 1; if (m > 64) m = 64; return m << 20; }();
    return v;
}
inline int pin_threads() {
    static const int v = [] { const char* e = getenv("GPSIG_PIN_THREADS"); int t = e ? atoi(e) : 4; if (t < 1) t = 1; if (t > PIN_THREADS_MAX) t = PIN_THREADS_MAX; return t; }();
    return v;
}
#define PIN_CHUNK (pin_chunk())
#define PIN_THREADS (pin_threads())

inline void host_copy_threaded(void* dst, const void* src, size_t bytes) {
    if (bytes < (size_t(1) << 20)) { memcpy(dst, src, bytes); return; }
    const size_t part = (bytes / PIN_THREADS + 4095) / 4096 * 4096;
    std::thread th[PIN_THREADS_MAX];
    int nt = 0;
    for (int k = 1; k < PIN_THREADS; ++k) {
        const size_t o = part * k;
        if (o >= bytes) break;
        const size_t n = bytes - o < part ? bytes - o : part;
        th[nt++] = std::thread([=] { memcpy(static_cast<char*>(dst) + o, static_cast<const char*>(src) + o, n); });
    }
    memcpy(dst, src, part < bytes ? part : bytes);
    for (int k = 0; k < nt; ++k) th[k].join();
}

// is `p` page-locked host memory HIP knows (hipHostMalloc / hipHostRegister, e.g. a torch pinned tensor)?  Then the DMA engines read
// and write it directly at full speed and no bounce buffer is needed.
inline bool host_is_pinned(const void* p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

inline int pin_ready(gpsig_ctx* c) {
    if (c->pin[0]) return GPSIG_OK;
    CHK(no_capture(c, "pinned staging buffers have to be allocated"));
    for (int k = 0; k < 2; ++k) {
        hipError_t e = hipHostMalloc(&c->pin[k], PIN_CHUNK_MAX, hipHostMallocDefault);
        if (e != hipSuccess) { c->pin[k] = nullptr; return fail(c, GPSIG_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", PIN_CHUNK, hipGetErrorString(e)); }
        HIPCHK(c, hipEventCreateWithFlags(&c->pin_ev[k], hipEventDisableTiming));
    }
    c->pin_bytes = PIN_CHUNK;
    return GPSIG_OK;
}

// host -> device on the ctx stream; returns once `user` has been read (the device copy completes in stream order)
inline int staged_h2d(gpsig_ctx* c, void* dev, const void* user, size_t bytes) {
    if (!c->pinned_staging || bytes < PIN_MIN || c->capturing || host_is_pinned(user)) {
        HIPCHK(c, hipMemcpyAsync(dev, user, bytes, hipMemcpyHostToDevice, c->stream));
        return GPSIG_OK;
    }
    CHK(pin_ready(c));
    size_t off = 0;
    for (int k = 0; off < bytes; ++k, off += PIN_CHUNK) {
        const int b = k & 1;
        const size_t n = bytes - off < PIN_CHUNK ? bytes - off : PIN_CHUNK;
        if (k >= 2) HIPCHK(c, hipEventSynchronize(c->pin_ev[b]));            // the DMA that last read this buffer is done
        host_copy_threaded(c->pin[b], static_cast<const char*>(user) + off, n);
        HIPCHK(c, hipMemcpyAsync(static_cast<char*>(dev) + off, c->pin[b], n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->pin_ev[b], c->stream));
    }
    // both buffers may be reused by the next transfer only after their DMAs: wait here (the data is on its way; kernels queued
    // behind it on the stream start as soon as it lands)
    HIPCHK(c, hipEventSynchronize(c->pin_ev[0]));
    HIPCHK(c, hipEventSynchronize(c->pin_ev[1]));
    return GPSIG_OK;
}

// device -> host, ordered after everything queued on the ctx stream; returns when `user` holds the data
inline int staged_d2h(gpsig_ctx* c, void* user, const void* dev, size_t bytes) {
    if (!c->pinned_staging || bytes < PIN_MIN || c->capturing || host_is_pinned(user)) {
        HIPCHK(c, hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, c->stream));
        return GPSIG_OK;
    }
    CHK(pin_ready(c));
    const size_t nchunks = (bytes + PIN_CHUNK - 1) / PIN_CHUNK;
    for (size_t k = 0; k <= nchunks; ++k) {
        if (k < nchunks) {
            const size_t off = k * PIN_CHUNK, n = bytes - off < PIN_CHUNK ? bytes - off : PIN_CHUNK;
            HIPCHK(c, hipMemcpyAsync(c->pin[k & 1], static_cast<const char*>(dev) + off, n, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipEventRecord(c->pin_ev[k & 1], c->stream));
        }
        if (k >= 1) {                                                         // chunk k-1 has landed: out of the bounce buffer while chunk k flies
            const size_t off = (k - 1) * PIN_CHUNK, n = bytes - off < PIN_CHUNK ? bytes - off : PIN_CHUNK;
            HIPCHK(c, hipEventSynchronize(c->pin_ev[(k - 1) & 1]));
            host_copy_threaded(static_cast<char*>(user) + off, c->pin[(k - 1) & 1], n);
        }
    }
    return GPSIG_OK;
}

// input pointer as the caller gave it -> device pointer
inline int in_dev(gpsig_ctx* c, int id, const void* user, size_t bytes, const void** dev) {
    if (!user && bytes) return fail(c, GPSIG_ERR_INVALID, "null input pointer");
    if (c->ptr_mode == GPSIG_PTR_DEVICE && user) { *dev = user; return GPSIG_OK; }
    void* p;
    CHK(ensure(c, id, bytes ? bytes : 8, &p));
    if (bytes && c->ptr_mode == GPSIG_PTR_HOST) CHK(staged_h2d(c, p, user, bytes));
    *dev = p;
    return GPSIG_OK;
}
inline int out_dev(gpsig_ctx* c, int id, void* user, size_t bytes, void** dev) {
    if (!user && bytes) return fail(c, GPSIG_ERR_INVALID, "null output pointer");
    if (c->ptr_mode == GPSIG_PTR_DEVICE && user) { *dev = user; return GPSIG_OK; }
    return ensure(c, id, bytes ? bytes : 8, dev);
}
inline int out_done(gpsig_ctx* c, void* user, const void* dev, size_t bytes) {
    if (c->ptr_mode == GPSIG_PTR_DEVICE || !user) return GPSIG_OK;
    if (bytes) CHK(staged_d2h(c, user, dev, bytes));
    return GPSIG_OK;
}
inline int finish(gpsig_ctx* c) {
    if (c->ptr_mode == GPSIG_PTR_HOST) CHK(host_sync(c));
    return GPSIG_OK;
}

inline int grid_for(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return int(g);
}

