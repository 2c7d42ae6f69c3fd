// wide_api.hip -- host side of the wide-state-space route (wide_kernels.hpp): augmented rows, the kernel-argument array by rocBLAS dgemm in chunks
// of whole sequences, the fused map / difference / chain kernels, and for the reverse pass the adjoint array contracted back by two more dgemms.
// Called from api.hip (evaluations, level primitives) and grad_api.hip (their gradients) with device pointers; *done = false leaves the call to the
// exact-shape kernels.  float64, order 1, RBF and the Matern families (the distance kernels: kappa is a function of the ONE number the dgemm yields).
#include "ctx.hpp"
#include "aux_kernels.hpp"
#include "wide_kernels.hpp"

#include <string>


namespace gpsig {

bool solver_dgemm(void** handle_slot, hipStream_t stream, bool transA, bool transB, int m, int n, int k, double alpha, const double* A, int lda,
                  const double* B, int ldb, double beta, double* C, int ldc, std::string* err);          // lowrank_solver.hip

namespace {

bool wide_kind(int base_kernel) {
    return base_kernel == GPSIG_BASE_RBF || base_kernel == GPSIG_BASE_MATERN12 || base_kernel == GPSIG_BASE_MATERN32 || base_kernel == GPSIG_BASE_MATERN52;
}

size_t wide_chunk_bytes(const gpsig_ctx* c) {
    if (c->wide_chunk_mb > 0) return size_t(c->wide_chunk_mb) << 20;
    return (size_t(c->grad_scratch_mb > 0 ? c->grad_scratch_mb : 4096) << 20) / 4;       // a quarter of the gradient path's scratch budget
}

int dgemm(gpsig_ctx* c, bool ta, bool tb, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B, int64_t ldb, double beta,
          double* C, int64_t ldc) {
    if (m > 0x7fffffff || n > 0x7fffffff || k > 0x7fffffff || lda > 0x7fffffff || ldb > 0x7fffffff || ldc > 0x7fffffff)
        return fail(c, GPSIG_ERR_UNSUPPORTED, "wide route: a matrix dimension beyond 2^31");
    if (m == 0 || n == 0) return GPSIG_OK;
    std::string err;
    if (!solver_dgemm(&c->blas_handle, c->stream, ta, tb, int(m), int(n), int(k), 1.0, A, int(lda), B, int(ldb), beta, C, int(ldc), &err))
        return fail(c, GPSIG_ERR_HIP, "%s", err.c_str());
    return GPSIG_OK;
}

// the augmented rows of both sides: ZA (lt * E * Tpad, DA) left form, XA (N * L, DA) right form
int wide_tvs_rows(gpsig_ctx* c, const ScaleParams& sz, const double* Z, const double* Xs, int lt, int E, int64_t Tn, int64_t Tpad, int64_t NL, int d,
                  double** ZA, double** XA) {
    const int DA = d + 2;
    const int64_t zr = int64_t(lt) * E * Tpad;
    void *za, *xa;
    CHK(ensure(c, B_WD0, sizeof(double) * size_t(zr) * DA + 64, &za));
    CHK(ensure(c, B_WD1, sizeof(double) * size_t(NL) * DA + 64, &xa));
    ScaleParams none;
    memset(&none, 0, sizeof(none));
    none.d_in = d;
    hipLaunchKernelGGL(wide_aug_rows_kernel, dim3(unsigned(zr < 65535 ? zr : 65535)), dim3(64), 0, c->stream, Z, zr, d, 0, lt, Tn, Tpad, E, sz,
                       static_cast<double*>(za));
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(wide_aug_rows_kernel, dim3(unsigned(NL < 65535 ? NL : 65535)), dim3(64), 0, c->stream, Xs, NL, d, 1, 0, int64_t(0), int64_t(0), 1,
                       none, static_cast<double*>(xa));
    HIPCHK(c, hipGetLastError());
    *ZA = static_cast<double*>(za); *XA = static_cast<double*>(xa);
    return GPSIG_OK;
}

// a pair of events from the context's pool around the timed launches (gpsig_timing_*; as api.hip: timing_begin_any)
int wide_timing_begin(gpsig_ctx* c, hipEvent_t* e0, hipEvent_t* e1, bool* on) {
    *on = false;
    if (c->capturing || c->ev_used + 2 > 8192) return GPSIG_OK;
    if (c->ev_used + 2 > c->ev.size()) {
        hipEvent_t a, b;
        HIPCHK(c, hipEventCreate(&a));
        HIPCHK(c, hipEventCreate(&b));
        c->ev.push_back(a);
        c->ev.push_back(b);
    }
    *e0 = c->ev[c->ev_used];
    *e1 = c->ev[c->ev_used + 1];
    c->ev_used += 2;
    HIPCHK(c, hipEventRecord(*e0, c->stream));
    *on = true;
    return GPSIG_OK;
}

}  // namespace

// Is the route built for this call?  (float64 is the caller's business.)
bool wide_tvs_available(const gpsig_ctx* c, const gpsig_params* p, int d, int64_t Tn, int64_t N, int L) {
    if (c->wide == 0 || c->capturing) return false;
    if (!wide_kind(p->base_kernel) || (p->order > 1 && p->num_levels > 1) || p->num_levels > WIDE_MAX_LEVELS || p->num_levels < 1) return false;
    if (Tn < 1 || N < 1 || L < 1 || d < 1) return false;
    return true;
}

// Kzx (kernels.py:313-340 + signature_algs.py:101-127 + the epilogue of kernels.py:572-588).  Z: the caller's (lt, T, E, d) array, scaled here when
// sz.has_ls; Xs: (N, L, d) scaled sequences.  fx (N, M+1), w (M+1): factors or NULL.  out: (T, N) level sum or (M+1, T, N).  aux: chain totals or NULL.
int wide_tvs_forward(gpsig_ctx* c, const gpsig_params* p, const ScaleParams& sz, int d, const double* Z, const double* Xs, int64_t Tn, int64_t N, int L,
                     int increments, const double* fx, const double* w, int sum_levels, double* out, double* aux) {
    const int M = p->num_levels, lt = M * (M + 1) / 2, E = increments ? 2 : 1, DA = d + 2;
    const int64_t Tpad = (Tn + 63) / 64 * 64, CW = int64_t(lt) * E * Tpad, TB = Tpad / 64;
    double *ZA, *XA;
    CHK(wide_tvs_rows(c, sz, Z, Xs, lt, E, Tn, Tpad, N * int64_t(L), d, &ZA, &XA));
    const size_t per_seq = sizeof(double) * size_t(L) * CW;
    int64_t chunk = int64_t(wide_chunk_bytes(c) / per_seq);
    if (chunk < 1) chunk = 1;
    if (chunk > N) chunk = N;
    void* arg;
    CHK(ensure(c, B_WD2, per_seq * size_t(chunk) + 64, &arg));
    hipEvent_t e0, e1;
    bool timed;
    CHK(wide_timing_begin(c, &e0, &e1, &timed));
    for (int64_t n0 = 0; n0 < N; n0 += chunk) {
        const int64_t nc = N - n0 < chunk ? N - n0 : chunk;
        // row-major arg (nc L, CW) = XA_chunk (nc L, DA) ZA^T  ==  column-major arg^T (CW x nc L) = ZA_cm^T (CW x DA) XA_cm (DA x nc L)
        CHK(dgemm(c, true, false, CW, nc * L, DA, ZA, DA, XA + n0 * L * DA, DA, 0.0, static_cast<double*>(arg), CW));
        WideTvsArgs A;
        memset(&A, 0, sizeof(A));
        A.arg = static_cast<const double*>(arg); A.CW = CW; A.Tpad = Tpad; A.Tn = Tn; A.n0 = n0; A.Nc = nc; A.N = N;
        A.L = L; A.M = M; A.kind = p->base_kernel; A.difference = p->difference ? 1 : 0; A.sum_levels = sum_levels;
        A.fx = fx; A.w = w; A.out = out; A.aux = aux;
        const dim3 grid(unsigned(TB), unsigned(nc < 65535 ? nc : 65535));
        if (E == 2) hipLaunchKernelGGL(wide_tvs_fwd_kernel<2>, grid, dim3(64), 0, c->stream, A);
        else hipLaunchKernelGGL(wide_tvs_fwd_kernel<1>, grid, dim3(64), 0, c->stream, A);
        HIPCHK(c, hipGetLastError());
    }
    if (timed) {
        HIPCHK(c, hipEventRecord(e1, c->stream));
        c->t_launches += 1;
        c->t_pairs += Tn * N;
        c->t_kernel = "wide_tvs (dgemm + wide_tvs_fwd_kernel)";
        c->t_flops += 2.0 * double(CW) * double(N) * L * DA;
    }
    return GPSIG_OK;
}

// The reverse pass.  Z (lt, T, E, d), X (N, L, d): scaled operands as the gradient entry points take them.  fac == NULL: G (M+1, T, N), gradient of the
// level array; fac (N, M+1): G (T, N), gradient of the weighted level sum, and gfac (N, M+1) receives the factors' gradient.
int wide_tvs_backward(gpsig_ctx* c, const gpsig_params* p, int d, const double* Z, const double* X, const double* G, int64_t Tn, int64_t N, int L,
                      int increments, const double* fac, const double* aux, double* gZ, double* gX, double* gfac) {
    const int M = p->num_levels, lt = M * (M + 1) / 2, E = increments ? 2 : 1, DA = d + 2;
    const int64_t Tpad = (Tn + 63) / 64 * 64, CW = int64_t(lt) * E * Tpad, TB = Tpad / 64, NL = N * int64_t(L);
    ScaleParams none;
    memset(&none, 0, sizeof(none));
    none.d_in = d;
    double *ZA, *XA;
    CHK(wide_tvs_rows(c, none, Z, X, lt, E, Tn, Tpad, NL, d, &ZA, &XA));
    const size_t per_seq = sizeof(double) * size_t(L) * CW;
    int64_t chunk = int64_t(wide_chunk_bytes(c) / per_seq);
    if (chunk < 1) chunk = 1;
    if (chunk > N) chunk = N;
    void *arg, *Wb, *gza, *gxa, *gfp = nullptr;
    CHK(ensure(c, B_WD2, per_seq * size_t(chunk) + 64, &arg));
    CHK(ensure(c, B_WD3, per_seq * size_t(chunk) + 64, &Wb));
    CHK(ensure(c, B_WD4, sizeof(double) * size_t(CW) * DA + 64, &gza));
    CHK(ensure(c, B_WD5, sizeof(double) * size_t(NL) * DA + 64, &gxa));
    if (fac) CHK(ensure(c, B_WD6, sizeof(double) * size_t(TB) * N * (M + 1) + 64, &gfp));
    for (int64_t n0 = 0; n0 < N; n0 += chunk) {
        const int64_t nc = N - n0 < chunk ? N - n0 : chunk;
        CHK(dgemm(c, true, false, CW, nc * L, DA, ZA, DA, XA + n0 * L * DA, DA, 0.0, static_cast<double*>(arg), CW));
        WideTvsArgs A;
        memset(&A, 0, sizeof(A));
        A.arg = static_cast<const double*>(arg); A.CW = CW; A.Tpad = Tpad; A.Tn = Tn; A.n0 = n0; A.Nc = nc; A.N = N;
        A.L = L; A.M = M; A.kind = p->base_kernel; A.difference = p->difference ? 1 : 0;
        A.fx = fac; A.w = nullptr; A.aux = const_cast<double*>(aux);
        A.G = G; A.W = static_cast<double*>(Wb); A.gfac_part = static_cast<double*>(gfp); A.weighted = fac ? 1 : 0;
        const dim3 grid(unsigned(TB), unsigned(nc < 65535 ? nc : 65535));
        if (E == 2) hipLaunchKernelGGL(wide_tvs_bwd_kernel<2>, grid, dim3(64), 0, c->stream, A);
        else hipLaunchKernelGGL(wide_tvs_bwd_kernel<1>, grid, dim3(64), 0, c->stream, A);
        HIPCHK(c, hipGetLastError());
        // gZA (CW, DA) += W^T XA_chunk:  column-major gZA^T (DA x CW) = XA_cm (DA x nc L) W_cm^T (nc L x CW)
        CHK(dgemm(c, false, true, DA, CW, nc * L, XA + n0 * L * DA, DA, static_cast<const double*>(Wb), CW, n0 > 0 ? 1.0 : 0.0, static_cast<double*>(gza), DA));
        // gXA_chunk (nc L, DA) = W ZA:   column-major gXA^T (DA x nc L) = ZA_cm (DA x CW) W_cm (CW x nc L)
        CHK(dgemm(c, false, false, DA, nc * L, CW, ZA, DA, static_cast<const double*>(Wb), CW, 0.0, static_cast<double*>(gxa) + n0 * L * DA, DA));
    }
    const int64_t zrows = int64_t(lt) * Tn * E;
    hipLaunchKernelGGL(wide_unaug_rows_kernel, dim3(grid_for(zrows * d)), dim3(256), 0, c->stream, static_cast<const double*>(gza), ZA, zrows, d, 0, lt, Tn,
                       Tpad, E, gZ);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(wide_unaug_rows_kernel, dim3(grid_for(NL * d)), dim3(256), 0, c->stream, static_cast<const double*>(gxa), XA, NL, d, 1, 0, int64_t(1),
                       int64_t(0), 1, gX);
    HIPCHK(c, hipGetLastError());
    if (fac) {
        hipLaunchKernelGGL(wide_gfac_reduce_kernel, dim3(grid_for(N * (M + 1))), dim3(256), 0, c->stream, static_cast<const double*>(gfp), int(TB),
                           N * int64_t(M + 1), gfac);
        HIPCHK(c, hipGetLastError());
    }
    return GPSIG_OK;
}

}  // namespace gpsig
