// ctx.hpp -- the gpsig_ctx object and the host-side helpers shared by the translation units that implement
// the C ABI (api.hip: evaluation, grad_api.hip: gradients).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
    B_LR0, B_LR1, B_LR2, B_LR3, B_LR4, B_LR5, B_LR6, B_LR7, B_LR8,
    B_GR0, B_GR1, B_GR2, B_GR3, B_GR4, B_GR5, B_GR6, B_GR7,   // gradient path scratch
    B_SPEC,                                                    // spectral base-kernel table
    B_TASKS_DIAG, B_TASKS_W2A, B_TASKS_W2B,                    // task lists that stay valid across calls (see TaskCache)
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

struct gpsig_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int ptr_mode = GPSIG_PTR_HOST;
    int shard_i = 0, shard_n = 1;
    int use_glds = 1;
    int allow_exact = 1;
    int max_run = 0;
    int tens_lanes = -1;   // -1 auto, 0 sequence lanes, 1 tensor lanes
    int grad_scratch_mb = 4096;   // lattice scratch of one gradient launch
    int grad_impl = 0;            // 0: planner's choice, 1: one pair per thread + stored lattice, 2: one pair per thread scratch-free (tensor-vs-seq),
                                  // 3: wavefront kernel + stored lattice, 4: scratch-free wavefront kernel wherever it is built
    int tvs_zreg = -1;            // tensor-lane gradient: components in registers (1) or LDS (0); -1 = planner's choice
    std::string err;
    DevBuf buf[B_COUNT];
    std::vector<gpsig::SeqTask> host_tasks;
    TaskCache tc_main, tc_diag, tc_w2a, tc_w2b;
    int w2_flip = 0;
    std::vector<double> last_weights;      // what B_W currently holds
    // timing of the pair-recursion launches
    std::vector<hipEvent_t> ev;     // pairs (start, stop)
    size_t ev_used = 0;
    int64_t t_launches = 0, t_pairs = 0;
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

inline int ensure(gpsig_ctx* c, int id, size_t bytes, void** out) {
    DevBuf& b = c->buf[id];
    if (bytes > b.cap) {
        if (b.p) {
            HIPCHK(c, hipStreamSynchronize(c->stream));   // nothing in flight may still use the old block
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

// input pointer as the caller gave it -> device pointer
inline int in_dev(gpsig_ctx* c, int id, const void* user, size_t bytes, const void** dev) {
    if (!user && bytes) return fail(c, GPSIG_ERR_INVALID, "null input pointer");
    if (c->ptr_mode == GPSIG_PTR_DEVICE && user) { *dev = user; return GPSIG_OK; }
    void* p;
    CHK(ensure(c, id, bytes ? bytes : 8, &p));
    if (bytes && c->ptr_mode == GPSIG_PTR_HOST) HIPCHK(c, hipMemcpyAsync(p, user, bytes, hipMemcpyHostToDevice, c->stream));
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
    if (bytes) HIPCHK(c, hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    return GPSIG_OK;
}
inline int finish(gpsig_ctx* c) {
    if (c->ptr_mode == GPSIG_PTR_HOST) HIPCHK(c, hipStreamSynchronize(c->stream));
    return GPSIG_OK;
}

inline int grid_for(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return int(g);
}

