// lr_grad_api.hip -- low-rank sequence features with the landmarks and the whitening ON THE DEVICE, and their reverse pass (round 4).
//
// When the reference trains in low-rank mode, the Nystrom landmarks are gathered from the scaled inputs and their Gram is decomposed
// inside the differentiated graph (gpsig/low_rank_calculations.py:47-60): landmarks and whitening are functions of the trainable
// parameters, new at every step, and live where the step runs.  gpsig_lr_seq_features (api.hip) takes them from the host (or from a
// gpsig_lr_draw state) because the evaluation path's caller owns them; the training path's two entry points take device pointers:
//     gpsig_lr_seq_features_dev    Phi (N, F) = the fused feature kernel of lr_fused_kernel.hpp on (X, S, Wh)
//     gpsig_lr_seq_features_grad   dPhi (N, F) -> dX, dS, dWh, d base parameter: lr_grad_kernel.hpp
// The projections of an evaluation are value-independent random objects: they come from the host once per draw, are kept on the
// device by content (with the two transposed copies the reverse pass gathers over) and reused by every call that passes the same ones.
#include "ctx.hpp"
#include "lr_grad_kernel.hpp"

#include <algorithm>
#include <vector>

using namespace gpsig;

namespace gpsig {
int lr_fused_launch(hipStream_t stream, const LrFusedArgs& A, unsigned grid, int variant);
int lr_fused2_launch(hipStream_t stream, const LrFusedArgs& A, unsigned grid);
}

namespace {

uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 0x100000001b3ull; }
    for (; i < n; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
    return h;
}

// the projections of levels 2 .. M on the device: by output column, by first operand index, by second operand index
int upload_sketches(gpsig_ctx* c, int cc, int r, int nsk, const gpsig_sketch* sk, LrGradSketch* out) {
    if (nsk < 0 || nsk > LR_FUSED_MAX_SKETCHES) return fail(c, GPSIG_ERR_UNSUPPORTED, "low-rank mode is built for num_levels <= %d", LR_FUSED_MAX_SKETCHES + 1);
    if (nsk > 0 && !sk) return fail(c, GPSIG_ERR_INVALID, "NULL sketch array");
    uint64_t h = 0xcbf29ce484222325ull;
    int k2 = cc;
    size_t bytes = 64;
    for (int i = 0; i < nsk; ++i) {
        const gpsig_sketch& s = sk[i];
        if (s.k1 != cc || s.k2 != k2 || s.r != r || s.nnz < 0 || !s.colptr || (s.nnz > 0 && (!s.i1 || !s.i2 || !s.val)))
            return fail(c, GPSIG_ERR_INVALID, "sketch %d has shape (%d, %d) -> %d, expected (%d, %d) -> %d", i, s.k1, s.k2, s.r, cc, k2, r);
        const int64_t sd[4] = {s.k1, s.k2, s.r, s.nnz};
        h = fnv(h, sd, sizeof(sd));
        h = fnv(h, s.colptr, sizeof(int32_t) * (size_t(s.r) + 1));
        h = fnv(h, s.i1, sizeof(int32_t) * size_t(s.nnz));
        h = fnv(h, s.i2, sizeof(int32_t) * size_t(s.nnz));
        h = fnv(h, s.val, sizeof(double) * size_t(s.nnz));
        bytes += 3 * (sizeof(LrEntry) * (size_t(s.nnz) + 1) + 16) + sizeof(int32_t) * (size_t(s.r) + size_t(s.k1) + size_t(s.k2) + 3) + 48;
        k2 = r;
    }
    if (h == 0) h = 1;
    void* base;
    CHK(ensure(c, B_LR8, bytes, &base));
    const bool cached = c->lrg_hash == h && c->lrg_base == base && c->lrg_offsets.size() == size_t(6 * nsk);
    std::vector<unsigned char> host(cached ? 0 : bytes);
    std::vector<size_t> offs;
    size_t o = 0;
    auto place = [&](const void* src, size_t n) -> const unsigned char* {
        size_t at;
        if (cached) {
            at = c->lrg_offsets[offs.size()];
        } else {
            o = (o + 15) / 16 * 16;
            if (n) memcpy(host.data() + o, src, n);
            at = o;
            o += n;
        }
        offs.push_back(at);
        return static_cast<const unsigned char*>(base) + at;
    };
    for (int i = 0; i < nsk; ++i) {
        const gpsig_sketch& s = sk[i];
        std::vector<LrEntry> e0, e1, e2;
        std::vector<int32_t> p1, p2;
        if (!cached) {
            const size_t nnz = size_t(s.nnz);
            e0.resize(nnz + 1); e1.resize(nnz + 1); e2.resize(nnz + 1);
            p1.assign(size_t(s.k1) + 1, 0); p2.assign(size_t(s.k2) + 1, 0);
            for (size_t e = 0; e < nnz; ++e) {
                if (s.i1[e] < 0 || s.i1[e] >= s.k1 || s.i2[e] < 0 || s.i2[e] >= s.k2) return fail(c, GPSIG_ERR_INVALID, "sketch %d: entry %zu out of range", i, e);
                e0[e] = LrEntry{s.val[e], s.i1[e], s.i2[e]};
                ++p1[size_t(s.i1[e]) + 1]; ++p2[size_t(s.i2[e]) + 1];
            }
            for (int k = 0; k < s.k1; ++k) p1[size_t(k) + 1] += p1[size_t(k)];
            for (int k = 0; k < s.k2; ++k) p2[size_t(k) + 1] += p2[size_t(k)];
            std::vector<int32_t> n1(p1.begin(), p1.end() - 1), n2(p2.begin(), p2.end() - 1);
            for (int j = 0; j < s.r; ++j)                              // entries in their given order: a row's entries keep it
                for (int32_t e = s.colptr[j]; e < s.colptr[j + 1]; ++e) {
                    e1[size_t(n1[size_t(s.i1[e])]++)] = LrEntry{s.val[e], s.i2[e], j};
                    e2[size_t(n2[size_t(s.i2[e])]++)] = LrEntry{s.val[e], s.i1[e], j};
                }
        }
        LrGradSketch& g = out[i];
        g.colptr = reinterpret_cast<const int32_t*>(place(s.colptr, sizeof(int32_t) * (size_t(s.r) + 1)));
        g.ent = reinterpret_cast<const LrEntry*>(place(e0.data(), sizeof(LrEntry) * size_t(s.nnz)));
        g.ptr1 = reinterpret_cast<const int32_t*>(place(p1.data(), sizeof(int32_t) * (size_t(s.k1) + 1)));
        g.ent1 = reinterpret_cast<const LrEntry*>(place(e1.data(), sizeof(LrEntry) * size_t(s.nnz)));
        g.ptr2 = reinterpret_cast<const int32_t*>(place(p2.data(), sizeof(int32_t) * (size_t(s.k2) + 1)));
        g.ent2 = reinterpret_cast<const LrEntry*>(place(e2.data(), sizeof(LrEntry) * size_t(s.nnz)));
    }
    if (!cached) {
        CHK(no_capture(c, "the projections of a low-rank evaluation have to be uploaded"));
        ++c->alloc_gen;                  // a recorded graph read the old contents of this buffer
        c->lrg_hash = 0;
        HIPCHK(c, hipMemcpyAsync(base, host.data(), o, hipMemcpyHostToDevice, c->stream));
        CHK(host_sync(c));               // `host` goes out of scope
        c->lrg_hash = h; c->lrg_base = base; c->lrg_offsets = offs;
    }
    return GPSIG_OK;
}

int check(gpsig_ctx* c, const gpsig_params* p, int cc, int r, int nsk) {
    if (!c) return GPSIG_ERR_INVALID;
    if (!p) return fail(c, GPSIG_ERR_INVALID, "params is NULL");
    if (p->dtype != GPSIG_F64) return fail(c, GPSIG_ERR_UNSUPPORTED, "low-rank mode is built for float64 only");
    if (p->base_kernel == GPSIG_BASE_SPECTRAL) return fail(c, GPSIG_ERR_UNSUPPORTED, "low-rank mode is not built for the spectral base kernel");
    if (p->base_kernel < GPSIG_BASE_LINEAR || p->base_kernel > GPSIG_BASE_MATERN52) return fail(c, GPSIG_ERR_INVALID, "unknown base kernel %d", p->base_kernel);
    if (p->num_levels < 1) return fail(c, GPSIG_ERR_INVALID, "num_levels must be >= 1");
    if (p->order != 1 && p->num_levels > 1) return fail(c, GPSIG_ERR_UNSUPPORTED, "Low-rank mode not implemented for order higher than 1.");
    if (cc < 1 || r < 1) return fail(c, GPSIG_ERR_INVALID, "num_components and rank_bound must be positive");
    if (nsk != p->num_levels - 1) return fail(c, GPSIG_ERR_INVALID, "need one sketch per level 2..num_levels");
    if (c->ptr_mode != GPSIG_PTR_DEVICE) return fail(c, GPSIG_ERR_INVALID, "the training-path low-rank entry points take device pointers");
    if (p->num_features < 1 || p->num_lags != 0) return fail(c, GPSIG_ERR_INVALID, "the level primitives take their columns as they come (num_lags = 0)");
    HIPCHK(c, hipSetDevice(c->device));
    return GPSIG_OK;
}

}  // namespace

extern "C" {

int gpsig_lr_seq_features_dev(gpsig_ctx* c, const gpsig_params* p, int32_t cc, int32_t r, int32_t nsk, const gpsig_sketch* sketches, const void* X,
                              int64_t N, int32_t L, const double* S, const double* Wh, void* Phi) {
    CHK(check(c, p, cc, r, nsk));
    if (N < 0 || L < 1 || (N > 0 && (!X || !S || !Wh || !Phi))) return fail(c, GPSIG_ERR_INVALID, "bad sizes / NULL pointer");
    const int M = p->num_levels, d = p->num_features, F = 1 + cc + (M - 1) * r;
    const size_t lds = lr_fused_lds_bytes(cc, r, d, L, c->lr_fused_pad);
    if (lds > LR_FUSED_MAX_LDS) return fail(c, GPSIG_ERR_UNSUPPORTED, "a sequence's low-rank arrays (%zu bytes) exceed the LDS", lds);
    LrGradSketch gs[LR_FUSED_MAX_SKETCHES];
    CHK(upload_sketches(c, cc, r, nsk, sketches, gs));
    if (N == 0) return GPSIG_OK;
    LrFusedArgs A;
    memset(&A, 0, sizeof(A));
    A.X = static_cast<const double*>(X); A.N = N; A.L = L;
    A.P.d_in = d;                          // no lengthscales, no lags: the caller scaled the inputs
    A.S = S; A.Wh = Wh;
    A.c = cc; A.r = r; A.M = M; A.difference = p->difference; A.kind = int(p->base_kernel);
    A.p0 = p->base_params[0]; A.p1 = p->base_params[1];
    for (int i = 0; i < nsk; ++i) A.sk[i] = LrFusedSketch{gs[i].colptr, gs[i].ent};
    A.Phi = static_cast<double*>(Phi); A.F = F;
    A.lp = lr_fused_stride(L, c->lr_fused_pad);
    A.rows_b = std::max(std::max(cc, r), d);
    const unsigned grid = unsigned(N < (int64_t(1) << 20) ? N : (int64_t(1) << 20));
    const int rc = lr_fused_launch(c->stream, A, grid, c->lr_fused_variant);
    if (rc != 0) return fail(c, GPSIG_ERR_HIP, "fused low-rank feature kernel: %s", hipGetErrorString(hipError_t(rc)));
    return GPSIG_OK;
}

int gpsig_lr_seq_features_grad(gpsig_ctx* c, const gpsig_params* p, int32_t cc, int32_t r, int32_t nsk, const gpsig_sketch* sketches, const void* X,
                               int64_t N, int32_t L, const double* S, const double* Wh, const void* dPhi, void* gX, double* gS, double* gWh,
                               double* g_base) {
    CHK(check(c, p, cc, r, nsk));
    if (N < 0 || L < 1 || !gS || !gWh || (N > 0 && (!X || !S || !Wh || !dPhi || !gX))) return fail(c, GPSIG_ERR_INVALID, "bad sizes / NULL pointer");
    const int M = p->num_levels, d = p->num_features, F = 1 + cc + (M - 1) * r;
    if (cc > 64 || int64_t(cc) * d > int64_t(LR_GRAD_KS) * LR_GRAD_THREADS)
        return fail(c, GPSIG_ERR_UNSUPPORTED, "the low-rank reverse pass is built for num_components <= 64 and num_components x columns <= %d", LR_GRAD_KS * LR_GRAD_THREADS);
    const size_t lds = lr_grad_lds_bytes(cc, r, d, L, c->lr_fused_pad);
    if (lds > LR_FUSED_MAX_LDS) return fail(c, GPSIG_ERR_UNSUPPORTED, "a sequence's low-rank arrays (%zu bytes) exceed the LDS in the reverse pass", lds);
    LrGradSketch gs[LR_FUSED_MAX_SKETCHES];
    CHK(upload_sketches(c, cc, r, nsk, sketches, gs));
    const int64_t width = int64_t(cc) * d + int64_t(cc) * cc + 1;
    if (N == 0) {
        CHK(zero_async(c, gS, sizeof(double) * size_t(cc) * d));
        CHK(zero_async(c, gWh, sizeof(double) * size_t(cc) * cc));
        if (g_base) CHK(zero_async(c, g_base, sizeof(double)));
        return GPSIG_OK;
    }
    const int l = p->difference ? L - 1 : L;
    const unsigned grid = unsigned(N < 512 ? N : 512);                 // one workgroup per CU at these LDS sizes, two rounds' worth of them
    const int64_t escr_stride = (int64_t(cc) + int64_t(M > 2 ? M - 2 : 0) * r) * (l > 0 ? l : 1) + 8;
    void *part, *escr;
    CHK(ensure(c, B_GR0, sizeof(double) * size_t(grid) * size_t(width) + 64, &part));
    CHK(ensure(c, B_GR1, sizeof(double) * size_t(grid) * size_t(escr_stride) + 64, &escr));
    LrGradArgs A;
    memset(&A, 0, sizeof(A));
    A.X = static_cast<const double*>(X); A.N = N; A.L = L; A.d = d;
    A.S = S; A.Wh = Wh;
    A.c = cc; A.r = r; A.M = M; A.difference = p->difference; A.kind = int(p->base_kernel);
    A.p0 = p->base_params[0]; A.p1 = p->base_params[1];
    for (int i = 0; i < nsk; ++i) A.sk[i] = gs[i];
    A.dPhi = static_cast<const double*>(dPhi); A.F = F;
    A.gX = static_cast<double*>(gX);
    A.part = static_cast<double*>(part);
    A.escr = static_cast<double*>(escr); A.escr_stride = escr_stride;
    A.lp = lr_fused_stride(L, c->lr_fused_pad);
    A.rows_b = std::max(std::max(std::max(cc, r), d), 16);
    // one workgroup per CU at these LDS sizes: 1024 threads give the scalar loads of the projections' entries twice the wavefronts to hide behind
    const bool wide = c->lr_grad_threads != 512;
    // (set on every launch that needs it, like sig_feat_grad_launch: a process-wide cache of the granted size would be wrong on a second device
    // and racy between contexts)
    const void* kern = wide ? reinterpret_cast<const void*>(lr_seq_features_grad_kernel<1024>) : reinterpret_cast<const void*>(lr_seq_features_grad_kernel<512>);
    if (lds > 48 * 1024) HIPCHK(c, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    if (wide) hipLaunchKernelGGL(lr_seq_features_grad_kernel<1024>, dim3(grid), dim3(1024), lds, c->stream, A);
    else hipLaunchKernelGGL(lr_seq_features_grad_kernel<512>, dim3(grid), dim3(512), lds, c->stream, A);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(lr_grad_reduce_kernel, dim3(unsigned((width + 255) / 256)), dim3(256), 0, c->stream, static_cast<const double*>(part), int(grid), width,
                       gS, int64_t(cc) * d, gWh, int64_t(cc) * cc, g_base);
    HIPCHK(c, hipGetLastError());
    return GPSIG_OK;
}

}  // extern "C"
