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

bool solver_dgemm_batched(void** handle_slot, hipStream_t stream, bool transA, bool transB, int m, int n, int k, double alpha, const double* A, int lda,
                          int64_t sa, const double* B, int ldb, int64_t sb, double beta, double* C, int ldc, int64_t sc, int batch, std::string* err);

// grad_api.hip: the sweeps of the higher-order reverse pass (grad_wave_ho_kernel.hpp)
struct WaveHoArgs;
typedef hipError_t (*WaveHoLaunchFn)(const WaveHoArgs&, int, size_t, hipStream_t);
struct HoSweeps { WaveHoLaunchFn fn; int G, C; size_t lds, slot; };
bool ho_sweeps_plan(const gpsig_ctx* c, const gpsig_params* p, int R1, int R2, HoSweeps* hs);
int ho_sweeps_launch(gpsig_ctx* c, const HoSweeps& hs, int M, int R1, int R2, const double* dM, double* lam, const double* G, int64_t gm, int64_t gi,
                     int64_t gj, int64_t N2, bool diag, int64_t pair0, int64_t npairs);
bool ho_levels_plan(const gpsig_ctx* c, const gpsig_params* p, int R1, int R2, HoSweeps* hs);
bool o1_sweeps_plan(const gpsig_ctx* c, const gpsig_params* p, int R1, int R2, HoSweeps* hs);
int ho_levels_launch(gpsig_ctx* c, const HoSweeps& hs, int M, int R1, int R2, const double* dM, double* out, int64_t gm, int64_t gi, int64_t gj, int64_t N2,
                     bool diag, int64_t pair0, int64_t npairs);

namespace {

bool wide_kind(int base_kernel) {
    return base_kernel == GPSIG_BASE_RBF || base_kernel == GPSIG_BASE_MATERN12 || base_kernel == GPSIG_BASE_MATERN32 || base_kernel == GPSIG_BASE_MATERN52;
}

size_t wide_chunk_bytes(const gpsig_ctx* c) {
    if (c->wide_chunk_mb > 0) return size_t(c->wide_chunk_mb) << 20;
    // 4 GB by default (the forward pass holds one such array, the reverse pass two and the partial sums of the narrow contraction: a launch of a
    // minibatch is bound by the serial sweep of one chain, so halving it doubles that time -- NetFlow's shape is 2 GB); grad_scratch_mb scales it
    return size_t(c->grad_scratch_mb > 0 ? c->grad_scratch_mb : 4096) << 20;
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

int dgemm_batched(gpsig_ctx* c, bool ta, bool tb, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, int64_t sa, const double* B, int64_t ldb,
                  int64_t sb, double* C, int64_t ldc, int64_t sc, int64_t batch) {
    if (m > 0x7fffffff || n > 0x7fffffff || k > 0x7fffffff || batch > 0x7fffffff)
        return fail(c, GPSIG_ERR_UNSUPPORTED, "wide route: a matrix dimension beyond 2^31");
    if (m == 0 || n == 0 || batch == 0) return GPSIG_OK;
    std::string err;
    if (!solver_dgemm_batched(&c->blas_handle, c->stream, ta, tb, int(m), int(n), int(k), 1.0, A, int(lda), sa, B, int(ldb), sb, 0.0, C, int(ldc), sc, int(batch), &err))
        return fail(c, GPSIG_ERR_HIP, "%s", err.c_str());
    return GPSIG_OK;
}

int aug_rows(gpsig_ctx* c, const double* src, int64_t rows, int d, int right, double* dst) {
    if (rows == 0) return GPSIG_OK;
    ScaleParams none;
    memset(&none, 0, sizeof(none));
    none.d_in = d;
    hipLaunchKernelGGL(wide_aug_rows_kernel, dim3(unsigned(rows < 65535 ? rows : 65535)), dim3(64), 0, c->stream, src, rows, d, right, 0, int64_t(0), int64_t(0), 1,
                       none, dst);
    HIPCHK(c, hipGetLastError());
    return GPSIG_OK;
}

typedef void (*WideLatKernel)(const WideLatArgs);
template <int LQ, bool RBF>
WideLatKernel lat_kernel_of(int C, bool bwd, int NW) {
    if (NW == 2) return bwd ? wide_lattice_bwd_kernel<1, LQ, RBF, 2> : wide_lattice_fwd_kernel<1, LQ, RBF, 2>;
    if (NW == 4) return bwd ? wide_lattice_bwd_kernel<1, LQ, RBF, 4> : wide_lattice_fwd_kernel<1, LQ, RBF, 4>;
    if (NW == 8) return bwd ? wide_lattice_bwd_kernel<1, LQ, RBF, 8> : wide_lattice_fwd_kernel<1, LQ, RBF, 8>;
    switch (C) {
        case 1: return bwd ? wide_lattice_bwd_kernel<1, LQ, RBF> : wide_lattice_fwd_kernel<1, LQ, RBF>;
        case 2: return bwd ? wide_lattice_bwd_kernel<2, LQ, RBF> : wide_lattice_fwd_kernel<2, LQ, RBF>;
        case 4: return bwd ? wide_lattice_bwd_kernel<4, LQ, RBF> : wide_lattice_fwd_kernel<4, LQ, RBF>;
        default: return bwd ? wide_lattice_bwd_kernel<8, LQ, RBF> : wide_lattice_fwd_kernel<8, LQ, RBF>;
    }
}
// levels kept by a lane: 3 (num_levels <= 4) or 7; the RBF kernel at compile time, the Matern families at run time.  NW > 1: NW wavefronts per
// lattice with one column per lane (C is ignored)
WideLatKernel lat_kernel(int M, int C, bool bwd, bool rbf, int NW = 1) {
    if (M <= 4) return rbf ? lat_kernel_of<3, true>(C, bwd, NW) : lat_kernel_of<3, false>(C, bwd, NW);
    return rbf ? lat_kernel_of<7, true>(C, bwd, NW) : lat_kernel_of<7, false>(C, bwd, NW);
}
// wavefronts per lattice: a launch of few lattices of more than 256 columns spreads each lattice's columns over eight wavefronts of one column per
// lane instead of eight columns per lane of one (NetFlow's / CMUsubject16's level diagonals, 50 / 23 lattices of 499 x 499: reverse pass 3.5 -> 3.2 / 2.9 ms;
// at four and two columns per lane the barrier per step costs more than the shorter steps save: AUSLAN 0.74 -> 0.79).  Option wide_lat_waves: 0 never,
// 1 wherever there are two or more columns per lane (the tests), -1 this rule
int lat_waves(const gpsig_ctx* c, int C, int64_t lattices) {
    if (C < 2 || c->wide_lat_waves == 0) return 1;
    if (c->wide_lat_waves > 0) return C;
    return (C == 8 && lattices <= 128) ? C : 1;
}
int lat_columns(int R2) { return R2 <= 64 ? 1 : (R2 <= 128 ? 2 : (R2 <= 256 ? 4 : 8)); }

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
// The two contractions of a reverse pass with the adjoint array W (R, CW) of an argument array  arg = XA (R, DA) ZA^T (CW, DA):
//     gXA (R, DA) = W ZA   (overwritten),      gZA (CW, DA) (+)= W^T XA   (accumulated over the chunks of R).
int contract_both(gpsig_ctx* c, const double* W, const double* XA, const double* ZA, int64_t R, int64_t CW, int DA, bool accumulate, double* gXA, double* gZA) {
    if (DA <= 32 && c->wide_contract != 0) {
        // narrow rows: both contractions in one hand-written pass over W (wide_contract_kernel); rocBLAS spreads such skinny products over too few tiles
        const int64_t strips = (R + 63) / 64, ntiles = (CW + 63) / 64;
        const int DAP = DA <= 16 ? 16 : 32;
        int64_t groups = ((DAP == 16 && c->wide_contract != 2 ? 8192 : 4096) + strips - 1) / strips;            // several wavefronts per SIMD
        if (groups > ntiles) groups = ntiles;
        if (groups < 1) groups = 1;
        void *part, *gxp;
        CHK(ensure(c, B_WD9, sizeof(double) * size_t(strips) * CW * DAP + 64, &part));
        CHK(ensure(c, B_WD8, sizeof(double) * size_t(groups) * R * DAP + 64, &gxp));
        WideContractArgs K;
        memset(&K, 0, sizeof(K));
        K.W = W; K.XA = XA; K.ZA = ZA; K.R = R; K.CW = CW; K.DA = DA; K.groups = int(groups);
        K.gXA_part = static_cast<double*>(gxp); K.part = static_cast<double*>(part);
        const dim3 gridc(unsigned(strips < 65535 ? strips : 65535), unsigned(groups));
        // 1 (default): part tiles of 16 rows (8.3 KB of LDS per wavefront; 1.18 ms per 4 GB of W at 12 columns where the first form takes 3.14), 3: of 32 rows (1.53), 2: the first form
        if (DAP == 16 && c->wide_contract == 3) hipLaunchKernelGGL(wide_contract16_kernel<32>, gridc, dim3(64), 0, c->stream, K);
        else if (DAP == 16 && c->wide_contract != 2) hipLaunchKernelGGL(wide_contract16_kernel<16>, gridc, dim3(64), 0, c->stream, K);
        else if (DAP == 16) hipLaunchKernelGGL(wide_contract_kernel<16>, gridc, dim3(64), 0, c->stream, K);
        else hipLaunchKernelGGL(wide_contract_kernel<32>, gridc, dim3(64), 0, c->stream, K);
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(wide_contract_reduce_kernel, dim3(grid_for(CW * DA)), dim3(256), 0, c->stream, static_cast<const double*>(part), strips, CW, DA, DAP,
                           accumulate ? 1 : 0, gZA);
        hipLaunchKernelGGL(wide_contract_reduce_kernel, dim3(grid_for(R * DA)), dim3(256), 0, c->stream, static_cast<const double*>(gxp), groups, R, DA, DAP, 0, gXA);
        HIPCHK(c, hipGetLastError());
        return GPSIG_OK;
    }
    // gZA (CW, DA) += W^T XA:  column-major gZA^T (DA x CW) = XA_cm (DA x R) W_cm^T (R x CW)
    CHK(dgemm(c, false, true, DA, CW, R, XA, DA, W, CW, accumulate ? 1.0 : 0.0, gZA, DA));
    // gXA (R, DA) = W ZA:   column-major gXA^T (DA x R) = ZA_cm (DA x CW) W_cm (CW x R)
    return dgemm(c, false, false, DA, R, CW, ZA, DA, W, CW, 0.0, gXA, DA);
}

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
    if (!wide_kind(p->base_kernel) || (p->order > WIDE_MAX_ORDER && p->num_levels > WIDE_MAX_ORDER) || p->num_levels > WIDE_MAX_LEVELS || p->num_levels < 1) return false;
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
    if (!aux) {          // the chain totals the epilogue reads: the caller's array (kept for the reverse pass) or scratch
        void* ax;
        CHK(ensure(c, B_WD8, sizeof(double) * size_t(N) * lt * Tpad + 64, &ax));
        aux = static_cast<double*>(ax);
    }
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
        A.fx = fx; A.w = w; A.out = out; A.aux = aux; A.order = p->order < p->num_levels ? p->order : p->num_levels;
        const dim3 grid(unsigned(TB), unsigned(nc < 65535 ? nc : 65535), unsigned(M));
        const bool rbf = p->base_kernel == GPSIG_BASE_RBF;
        if (E == 2) { if (rbf) hipLaunchKernelGGL((wide_tvs_fwd_kernel<2, true>), grid, dim3(64), 0, c->stream, A); else hipLaunchKernelGGL((wide_tvs_fwd_kernel<2, false>), grid, dim3(64), 0, c->stream, A); }
        else { if (rbf) hipLaunchKernelGGL((wide_tvs_fwd_kernel<1, true>), grid, dim3(64), 0, c->stream, A); else hipLaunchKernelGGL((wide_tvs_fwd_kernel<1, false>), grid, dim3(64), 0, c->stream, A); }
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(wide_tvs_epilogue_kernel, dim3(grid_for(nc * Tn)), dim3(256), 0, c->stream, A);
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
        A.order = p->order < p->num_levels ? p->order : p->num_levels;
        const dim3 grid(unsigned(TB), unsigned(nc < 65535 ? nc : 65535), unsigned(M));
        const bool rbf = p->base_kernel == GPSIG_BASE_RBF;
        if (E == 2) { if (rbf) hipLaunchKernelGGL((wide_tvs_bwd_kernel<2, true>), grid, dim3(64), 0, c->stream, A); else hipLaunchKernelGGL((wide_tvs_bwd_kernel<2, false>), grid, dim3(64), 0, c->stream, A); }
        else { if (rbf) hipLaunchKernelGGL((wide_tvs_bwd_kernel<1, true>), grid, dim3(64), 0, c->stream, A); else hipLaunchKernelGGL((wide_tvs_bwd_kernel<1, false>), grid, dim3(64), 0, c->stream, A); }
        HIPCHK(c, hipGetLastError());
        CHK(contract_both(c, static_cast<const double*>(Wb), XA + n0 * L * DA, ZA, nc * int64_t(L), CW, DA, n0 > 0, static_cast<double*>(gxa) + n0 * L * DA,
                          static_cast<double*>(gza)));
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


// ---- sequence lattices ----------------------------------------------------------------------------------------------------------------------
constexpr int WIDE_LAT_MAX_COLS = 512;            // 64 lanes x 8 columns

bool wide_lat_available(const gpsig_ctx* c, const gpsig_params* p, int L1, int L2) {
    if (c->wide == 0 || c->capturing) return false;
    if (!wide_kind(p->base_kernel) || p->num_levels > WIDE_MAX_LEVELS || p->num_levels < 1) return false;
    const int dr = p->difference ? 1 : 0;
    if (p->order > 1 && p->num_levels > 1) {      // higher orders: the forward and the reverse sweeps of grad_wave_ho_kernel.hpp (<= 5 levels, orders <= 4)
        HoSweeps hf, hb;
        return L1 - dr >= 1 && L2 - dr >= 1 && ho_levels_plan(c, p, L1 - dr, L2 - dr, &hf) && ho_sweeps_plan(c, p, L1 - dr, L2 - dr, &hb);
    }
    return L1 >= 1 && L2 >= 1 && L2 - dr <= WIDE_LAT_MAX_COLS;
}

// the reverse pass of the HIGHER-ORDER recursion on this route: argument lattices by dgemm -> dM lattices (wide_lattice_dm_kernel) -> both sweeps of a
// pair in one wavefront (grad_wave_ho_kernel.hpp) -> the adjoint contracted back by dgemms; any number of columns
bool wide_lat_ho_available(const gpsig_ctx* c, const gpsig_params* p, int L1, int L2) {
    if (c->wide == 0 || c->capturing || !wide_kind(p->base_kernel) || !(p->order > 1 && p->num_levels > 1)) return false;
    const int dr = p->difference ? 1 : 0;
    HoSweeps hs;
    return L1 >= 1 && L2 >= 1 && ho_sweeps_plan(c, p, L1 - dr, L2 - dr, &hs);
}

namespace {

struct LatPlan {
    int DA, dr, R1, R2, C;
    int64_t P, Ptot, chunk_i, ld, si, sj, N2;      // chunk_i: left sequences per chunk (diag: sequences per chunk)
    double *XL, *XR;
};

// augmented rows (left form of the left sequences, right form of the right ones) and the chunking of the argument lattices
int lat_plan(gpsig_ctx* c, const gpsig_params* p, int d, const double* Xs, const double* Ys, int64_t N1, int64_t N2, int L1, int L2, bool diag, int bufs,
             LatPlan* pl) {
    pl->DA = d + 2; pl->dr = p->difference ? 1 : 0; pl->R1 = L1 - pl->dr; pl->R2 = L2 - pl->dr; pl->C = lat_columns(pl->R2);
    void *xl, *xr;
    CHK(ensure(c, B_WD0, sizeof(double) * size_t(N1) * L1 * pl->DA + 64, &xl));
    CHK(ensure(c, B_WD1, sizeof(double) * size_t(N2) * L2 * pl->DA + 64, &xr));
    CHK(aug_rows(c, Xs, N1 * int64_t(L1), d, 0, static_cast<double*>(xl)));
    CHK(aug_rows(c, Ys ? Ys : Xs, N2 * int64_t(L2), d, 1, static_cast<double*>(xr)));
    pl->XL = static_cast<double*>(xl); pl->XR = static_cast<double*>(xr);
    const size_t per_i = sizeof(double) * size_t(L1) * L2 * size_t(diag ? 1 : N2) * size_t(bufs);
    int64_t chunk = int64_t(wide_chunk_bytes(c) / (per_i ? per_i : 1));
    if (chunk < 1) chunk = 1;
    if (chunk > N1) chunk = N1;
    pl->chunk_i = chunk;
    pl->Ptot = diag ? N1 : N1 * N2;
    if (diag) { pl->ld = L2; pl->si = int64_t(L1) * L2; pl->sj = 0; pl->N2 = 1; }
    else { pl->ld = N2 * int64_t(L2); pl->si = int64_t(L1) * pl->ld; pl->sj = L2; pl->N2 = N2; }
    return GPSIG_OK;
}

// the argument lattices of the left sequences i0 .. i0 + ni - 1 into `arg`
// (j0 > 0: the right sequences j0 .. N2 - 1 only -- the symmetric Gram's reverse pass, pairs i <= j)
int lat_arguments(gpsig_ctx* c, const LatPlan& pl, int64_t i0, int64_t ni, int64_t N2, int L1, int L2, bool diag, double* arg, int64_t j0 = 0) {
    const int DA = pl.DA;
    if (diag)      // per sequence: row-major arg (L1, L2) = XL_n XR_n^T  ==  column-major (L2 x L1) = XR_n,cm^T (L2 x DA) XL_n,cm (DA x L1)
        return dgemm_batched(c, true, false, L2, L1, DA, pl.XR + i0 * L2 * DA, DA, int64_t(L2) * DA, pl.XL + i0 * L1 * DA, DA, int64_t(L1) * DA, arg, L2,
                             int64_t(L1) * L2, ni);
    // row-major arg (ni L1, N2 L2) = XL_chunk XR^T  ==  column-major (N2 L2 x ni L1) = XR_cm^T XL_chunk,cm
    return dgemm(c, true, false, (N2 - j0) * L2, ni * L1, DA, pl.XR + j0 * L2 * DA, DA, pl.XL + i0 * L1 * DA, DA, 0.0, arg, (N2 - j0) * L2);
}

}  // namespace

// Raw levels (M+1, P) of the lattices of pairs (i, j) -- P = N1 N2, pair index i N2 + j -- or (i, i) -- diag, P = N1.  Xs, Ys: scaled sequences
// (Ys == NULL: Xs on both sides).  signature_algs.py:8-35 on kernels.py:188-237's tensors.
int wide_lat_forward(gpsig_ctx* c, const gpsig_params* p, int d, const double* Xs, const double* Ys, int64_t N1, int64_t N2, int L1, int L2, bool diag,
                     double* out) {
    LatPlan pl;
    const bool ho = p->order > 1 && p->num_levels > 1;
    HoSweeps hs;
    if (ho && !ho_levels_plan(c, p, L1 - (p->difference ? 1 : 0), L2 - (p->difference ? 1 : 0), &hs))
        return fail(c, GPSIG_ERR_UNSUPPORTED, "no higher-order forward sweep for this shape");
    CHK(lat_plan(c, p, d, Xs, Ys, N1, N2, L1, L2, diag, ho ? 2 : 1, &pl));
    const int M = p->num_levels;
    void *arg, *dmat = nullptr;
    CHK(ensure(c, B_WD2, sizeof(double) * size_t(pl.chunk_i) * L1 * L2 * size_t(diag ? 1 : N2) + 64, &arg));
    if (ho) CHK(ensure(c, B_WD6, sizeof(double) * size_t(pl.chunk_i) * L1 * L2 * size_t(diag ? 1 : N2) + 64, &dmat));
    const int NW = ho ? 1 : lat_waves(c, pl.C, diag ? (N1 < pl.chunk_i ? N1 : pl.chunk_i) : (N1 < pl.chunk_i ? N1 : pl.chunk_i) * N2);
    WideLatKernel fn = ho ? nullptr : lat_kernel(M, pl.C, false, p->base_kernel == GPSIG_BASE_RBF, NW);
    hipEvent_t e0, e1;
    bool timed;
    CHK(wide_timing_begin(c, &e0, &e1, &timed));
    for (int64_t i0 = 0; i0 < N1; i0 += pl.chunk_i) {
        const int64_t ni = N1 - i0 < pl.chunk_i ? N1 - i0 : pl.chunk_i;
        CHK(lat_arguments(c, pl, i0, ni, N2, L1, L2, diag, static_cast<double*>(arg)));
        WideLatArgs A;
        memset(&A, 0, sizeof(A));
        A.arg = static_cast<const double*>(arg); A.ld = pl.ld; A.si = pl.si; A.sj = pl.sj; A.N2 = pl.N2;
        A.P = diag ? ni : ni * N2; A.p0 = 0; A.Ptot = pl.Ptot;
        A.L1 = L1; A.L2 = L2; A.M = M; A.kind = p->base_kernel; A.difference = pl.dr;
        A.out = out + (diag ? i0 : i0 * N2);              // (the kernel's pair index starts at 0 in this chunk's lattices)
        if (ho) {
            if (pl.R1 < 1 || pl.R2 < 1) return fail(c, GPSIG_ERR_UNSUPPORTED, "empty lattices");
            if (p->base_kernel == GPSIG_BASE_RBF)
                hipLaunchKernelGGL(wide_lattice_dm_kernel<true>, dim3(grid_for(A.P * int64_t(pl.R1) * pl.R2)), dim3(256), 0, c->stream, A, static_cast<double*>(dmat));
            else
                hipLaunchKernelGGL(wide_lattice_dm_kernel<false>, dim3(grid_for(A.P * int64_t(pl.R1) * pl.R2)), dim3(256), 0, c->stream, A, static_cast<double*>(dmat));
            HIPCHK(c, hipGetLastError());
            CHK(ho_levels_launch(c, hs, M, pl.R1, pl.R2, static_cast<const double*>(dmat), out, pl.Ptot, diag ? 1 : N2, diag ? 0 : 1, diag ? 1 : N2, diag,
                                 diag ? i0 : i0 * N2, A.P));
            continue;
        }
        hipLaunchKernelGGL(fn, dim3(unsigned(A.P < 65535 ? A.P : 65535)), dim3(64 * NW), 0, c->stream, A);
        HIPCHK(c, hipGetLastError());
    }
    if (timed) {
        HIPCHK(c, hipEventRecord(e1, c->stream));
        c->t_launches += 1;
        c->t_pairs += pl.Ptot;
        c->t_kernel = "wide_lattice (dgemm + wide_lattice_fwd_kernel)";
    }
    return GPSIG_OK;
}

// Gradients of sum_m G[m][pair] level_m[pair] with respect to the scaled sequences.  G: (M+1, P).  Ys == NULL: Xs on both sides and both sides'
// gradients land in gX (the symmetric Gram as the cross Gram of X with itself; the diagonal); else gX, gY.
int wide_lat_backward(gpsig_ctx* c, const gpsig_params* p, int d, const double* Xs, const double* Ys, int64_t N1, int64_t N2, int L1, int L2, bool diag,
                      const double* G, double* gX, double* gY) {
    LatPlan pl;
    const bool ho = p->order > 1 && p->num_levels > 1;
    HoSweeps hs;
    if (ho && !ho_sweeps_plan(c, p, L1 - (p->difference ? 1 : 0), L2 - (p->difference ? 1 : 0), &hs))
        return fail(c, GPSIG_ERR_UNSUPPORTED, "no higher-order sweeps for this shape");
    // first order, MANY SHORT lattices (a Gram of sequences of at most 65 observations): four lattices per wavefront from a dM lattice (seq_grad_wave_o1_kernel) instead of
    // one per 64-lane wavefront; few lattices (a minibatch's level diagonals) stay with the lattice kernels below.  Option wide_o1_sweeps: 0 never, 1 (default) from 1,024 lattices, 2 wherever the shape fits (tests)
    bool short_lat = false;
    if (!ho && c->wide_o1_sweeps != 0 && ((diag ? N1 : N1 * N2) >= 1024 || c->wide_o1_sweeps == 2))
        short_lat = o1_sweeps_plan(c, p, L1 - (p->difference ? 1 : 0), L2 - (p->difference ? 1 : 0), &hs);
    CHK(lat_plan(c, p, d, Xs, Ys, N1, N2, L1, L2, diag, (ho || short_lat) ? 3 : 2, &pl));
    // the symmetric Gram (one array on both sides): the pairs i <= j with the upstream gradient folded onto them -- half the lattices
    const bool fold = !diag && Ys == nullptr && N1 == N2 && L1 == L2 && c->wide_sym_fold != 0;
    if (fold) {
        // (a chunk of ni left sequences still sweeps its ni x ni square whole -- zeros below the diagonal: chunks of at most N / 16 keep that under 4 %)
        const int64_t cap = N1 / 16 > 8 ? N1 / 16 : 8;
        if (pl.chunk_i > cap) pl.chunk_i = cap;
        void* gs;
        CHK(ensure(c, B_WD10, sizeof(double) * size_t(p->num_levels + 1) * N1 * N1 + 64, &gs));
        hipLaunchKernelGGL(wide_sym_upstream_kernel, dim3(grid_for(int64_t(p->num_levels + 1) * N1 * N1)), dim3(256), 0, c->stream, G, N1, p->num_levels + 1,
                           static_cast<double*>(gs));
        HIPCHK(c, hipGetLastError());
        G = static_cast<const double*>(gs);
    }
    const int M = p->num_levels, DA = pl.DA;
    const int64_t per_i = int64_t(L1) * L2 * (diag ? 1 : N2);
    void *arg, *lam, *gxl, *gxr, *scr, *dmat = nullptr;
    if (ho || short_lat) CHK(ensure(c, B_WD6, sizeof(double) * size_t(pl.chunk_i) * per_i + 64, &dmat));
    CHK(ensure(c, B_WD2, sizeof(double) * size_t(pl.chunk_i) * per_i + 64, &arg));
    CHK(ensure(c, B_WD3, sizeof(double) * size_t(pl.chunk_i) * per_i + 64, &lam));
    CHK(ensure(c, B_WD4, sizeof(double) * size_t(N1) * L1 * DA + 64, &gxl));
    CHK(ensure(c, B_WD5, sizeof(double) * size_t(N2) * L2 * DA + 64, &gxr));
    const int64_t Pmax = diag ? pl.chunk_i : pl.chunk_i * N2;
    const int NW = (ho || short_lat) ? 1 : lat_waves(c, pl.C, Pmax);
    const int TF = pl.R1 + 64 * NW - 1;
    const size_t per_group = sizeof(double) * size_t(M > 1 ? M - 1 : 1) * TF * 64 * (NW > 1 ? NW : pl.C);
    int64_t groups = int64_t(wide_chunk_bytes(c) / per_group);
    if (groups < 1) groups = 1;
    if (groups > Pmax) groups = Pmax;
    if (groups > 4096) groups = 4096;
    if (ho || short_lat) groups = 1;      // (these sweeps bring their own slots, if any)
    CHK(ensure(c, B_WD7, per_group * size_t(groups) + 64, &scr));
    WideLatKernel fn = (ho || short_lat) ? nullptr : lat_kernel(M, pl.C, true, p->base_kernel == GPSIG_BASE_RBF, NW);
    for (int64_t i0 = 0; i0 < N1; i0 += pl.chunk_i) {
        const int64_t ni = N1 - i0 < pl.chunk_i ? N1 - i0 : pl.chunk_i;
        const int64_t j0 = fold ? i0 : 0, N2e = N2 - j0;              // right sequences of this chunk
        CHK(lat_arguments(c, pl, i0, ni, N2, L1, L2, diag, static_cast<double*>(arg), j0));
        WideLatArgs A;
        memset(&A, 0, sizeof(A));
        A.arg = static_cast<const double*>(arg); A.ld = pl.ld; A.si = pl.si; A.sj = pl.sj; A.N2 = pl.N2;
        if (!diag) { A.ld = N2e * int64_t(L2); A.si = int64_t(L1) * A.ld; A.N2 = N2e; }
        A.P = diag ? ni : ni * N2e; A.p0 = 0; A.Ptot = pl.Ptot;
        A.L1 = L1; A.L2 = L2; A.M = M; A.kind = p->base_kernel; A.difference = pl.dr;
        A.G = G + (diag ? i0 : i0 * N2 + j0);
        A.g_i = diag ? 1 : N2; A.g_j = diag ? 0 : 1;
        A.scratch = static_cast<double*>(scr); A.lam = static_cast<double*>(lam);
        const int64_t ng = A.P < groups ? A.P : groups;
        A.ngroups = int(ng);
        if ((ho || short_lat) && pl.R1 > 0 && pl.R2 > 0) {
            if (p->base_kernel == GPSIG_BASE_RBF)
                hipLaunchKernelGGL(wide_lattice_dm_kernel<true>, dim3(grid_for(A.P * int64_t(pl.R1) * pl.R2)), dim3(256), 0, c->stream, A, static_cast<double*>(dmat));
            else
                hipLaunchKernelGGL(wide_lattice_dm_kernel<false>, dim3(grid_for(A.P * int64_t(pl.R1) * pl.R2)), dim3(256), 0, c->stream, A, static_cast<double*>(dmat));
            HIPCHK(c, hipGetLastError());
            // (the sweeps address G by (i, j') = divmod(pair0 + pair, N2e): i absolute with pair0 = i0 N2e, j' relative to the chunk's first right sequence)
            CHK(ho_sweeps_launch(c, hs, M, pl.R1, pl.R2, static_cast<const double*>(dmat), static_cast<double*>(lam), G + j0, pl.Ptot, diag ? 1 : N2, diag ? 0 : 1,
                                 diag ? 1 : N2e, diag, diag ? i0 : i0 * N2e, A.P));
        } else if (pl.R1 > 0 && pl.R2 > 0) {
            hipLaunchKernelGGL(fn, dim3(unsigned(ng)), dim3(64 * NW), 0, c->stream, A);
            HIPCHK(c, hipGetLastError());
        }
        // the adjoint of the arguments, in place of the arguments
        if (p->base_kernel == GPSIG_BASE_RBF)
            hipLaunchKernelGGL(wide_lattice_adjoint_kernel<true>, dim3(grid_for(A.P * int64_t(L1) * L2)), dim3(256), 0, c->stream, A, static_cast<double*>(arg));
        else
            hipLaunchKernelGGL(wide_lattice_adjoint_kernel<false>, dim3(grid_for(A.P * int64_t(L1) * L2)), dim3(256), 0, c->stream, A, static_cast<double*>(arg));
        HIPCHK(c, hipGetLastError());
        const double* W = static_cast<const double*>(arg);
        double* gl = static_cast<double*>(gxl) + i0 * L1 * DA;
        if (diag) {
            // gXL_n (L1, DA) = W_n XR_n: column-major (DA x L1) = XR_n,cm (DA x L2) W_n,cm (L2 x L1);  gXR_n (L2, DA) = W_n^T XL_n: (DA x L2) = XL_n,cm (DA x L1) W_n,cm^T
            CHK(dgemm_batched(c, false, false, DA, L1, L2, pl.XR + i0 * L2 * DA, DA, int64_t(L2) * DA, W, L2, int64_t(L1) * L2, gl, DA, int64_t(L1) * DA, ni));
            CHK(dgemm_batched(c, false, true, DA, L2, L1, pl.XL + i0 * L1 * DA, DA, int64_t(L1) * DA, W, L2, int64_t(L1) * L2,
                              static_cast<double*>(gxr) + i0 * L2 * DA, DA, int64_t(L2) * DA, ni));
        } else {
            CHK(contract_both(c, W, pl.XL + i0 * L1 * DA, pl.XR + j0 * L2 * DA, ni * int64_t(L1), N2e * int64_t(L2), DA, i0 > 0, gl,
                              static_cast<double*>(gxr) + j0 * L2 * DA));
        }
    }
    // through the augmentation: left form into gX; right form into gX as well (one array on both sides) or into gY
    const int64_t xr_rows = N1 * int64_t(L1), yr_rows = N2 * int64_t(L2);
    const bool same = Ys == nullptr;
    hipLaunchKernelGGL(wide_unaug_pair_kernel, dim3(grid_for(xr_rows * d)), dim3(256), 0, c->stream, static_cast<const double*>(gxl), pl.XL,
                       same ? static_cast<const double*>(gxr) : nullptr, same ? pl.XR : nullptr, xr_rows, d, gX);
    HIPCHK(c, hipGetLastError());
    if (!same) {
        hipLaunchKernelGGL(wide_unaug_rows_kernel, dim3(grid_for(yr_rows * d)), dim3(256), 0, c->stream, static_cast<const double*>(gxr), pl.XR, yr_rows, d, 1, 0,
                           int64_t(1), int64_t(0), 1, gY);
        HIPCHK(c, hipGetLastError());
    }
    return GPSIG_OK;
}

// ---- inducing tensors vs inducing tensors ----------------------------------------------------------------------------------------------------
bool wide_tens_available(const gpsig_ctx* c, const gpsig_params* p, int64_t Tn) {
    if (c->wide == 0 || c->capturing) return false;
    if (!wide_kind(p->base_kernel) || p->num_levels > WIDE_MAX_LEVELS || p->num_levels < 1) return false;
    const int64_t Tpad = (Tn + 63) / 64 * 64;
    // the argument blocks of every component at once (10 components x 1,024^2 x 8 bytes = 84 MB at 500 tensors with increments)
    return Tn >= 1 && size_t(p->num_levels * (p->num_levels + 1) / 2) * size_t(4 * Tpad * Tpad) * sizeof(double) <= (size_t(2) << 30);
}

namespace {
// left- and right-form augmented rows of the tensors (B_WD0, B_WD6) and the argument blocks (B_WD2)
int tens_arguments(gpsig_ctx* c, const ScaleParams& sz, int d, const double* Z, int lt, int E, int64_t Tn, int64_t Tpad, double** ZL, double** ZR, double** arg) {
    const int DA = d + 2;
    const int64_t zr = int64_t(lt) * E * Tpad, R = int64_t(E) * Tpad;
    void *zl, *zrr, *ar;
    CHK(ensure(c, B_WD0, sizeof(double) * size_t(zr) * DA + 64, &zl));
    CHK(ensure(c, B_WD6, sizeof(double) * size_t(zr) * DA + 64, &zrr));
    CHK(ensure(c, B_WD2, sizeof(double) * size_t(lt) * R * R + 64, &ar));
    for (int right = 0; right < 2; ++right) {
        hipLaunchKernelGGL(wide_aug_rows_kernel, dim3(unsigned(zr < 65535 ? zr : 65535)), dim3(64), 0, c->stream, Z, zr, d, right, lt, Tn, Tpad, E, sz,
                           static_cast<double*>(right ? zrr : zl));
        HIPCHK(c, hipGetLastError());
    }
    // block k, row-major (R, R) = ZL_k ZR_k^T  ==  column-major (R x R) = ZR_k,cm^T (R x DA) ZL_k,cm (DA x R)
    CHK(dgemm_batched(c, true, false, R, R, DA, static_cast<const double*>(zrr), DA, R * DA, static_cast<const double*>(zl), DA, R * DA, static_cast<double*>(ar), R,
                      R * R, lt));
    *ZL = static_cast<double*>(zl); *ZR = static_cast<double*>(zrr); *arg = static_cast<double*>(ar);
    return GPSIG_OK;
}
}  // namespace

// Kzz: Z the caller's (lt, T, E, d) array (scaled here when sz.has_ls); w (M+1) or NULL; out (T, T) weighted sum or (M+1, T, T)
int wide_tens_forward(gpsig_ctx* c, const gpsig_params* p, const ScaleParams& sz, int d, const double* Z, int64_t Tn, int increments, const double* w,
                      int sum_levels, double* out) {
    const int M = p->num_levels, lt = M * (M + 1) / 2, E = increments ? 2 : 1;
    const int64_t Tpad = (Tn + 63) / 64 * 64;
    double *ZL, *ZR, *arg;
    CHK(tens_arguments(c, sz, d, Z, lt, E, Tn, Tpad, &ZL, &ZR, &arg));
    WideTensArgs A;
    memset(&A, 0, sizeof(A));
    A.arg = arg; A.Tpad = Tpad; A.Tn = Tn; A.M = M; A.E = E; A.kind = p->base_kernel; A.sum_levels = sum_levels; A.w = w; A.out = out;
    if (p->base_kernel == GPSIG_BASE_RBF) hipLaunchKernelGGL(wide_tens_fwd_kernel<true>, dim3(unsigned(Tpad / 64), unsigned(Tn < 65535 ? Tn : 65535)), dim3(64), 0, c->stream, A);
    else hipLaunchKernelGGL(wide_tens_fwd_kernel<false>, dim3(unsigned(Tpad / 64), unsigned(Tn < 65535 ? Tn : 65535)), dim3(64), 0, c->stream, A);
    HIPCHK(c, hipGetLastError());
    return GPSIG_OK;
}

// gradient of sum_m G[m][t][t'] level_m[t][t'] with respect to the scaled tensors Z (lt, T, E, d)
int wide_tens_backward(gpsig_ctx* c, const gpsig_params* p, int d, const double* Z, int64_t Tn, int increments, const double* G, double* gZ) {
    const int M = p->num_levels, lt = M * (M + 1) / 2, E = increments ? 2 : 1, DA = d + 2;
    const int64_t Tpad = (Tn + 63) / 64 * 64, R = int64_t(E) * Tpad, zr = int64_t(lt) * R;
    ScaleParams none;
    memset(&none, 0, sizeof(none));
    none.d_in = d;
    double *ZL, *ZR, *arg;
    CHK(tens_arguments(c, none, d, Z, lt, E, Tn, Tpad, &ZL, &ZR, &arg));
    void *Wb, *gzl, *gzr;
    CHK(ensure(c, B_WD3, sizeof(double) * size_t(lt) * R * R + 64, &Wb));
    CHK(ensure(c, B_WD4, sizeof(double) * size_t(zr) * DA + 64, &gzl));
    CHK(ensure(c, B_WD5, sizeof(double) * size_t(zr) * DA + 64, &gzr));
    WideTensArgs A;
    memset(&A, 0, sizeof(A));
    A.arg = arg; A.Tpad = Tpad; A.Tn = Tn; A.M = M; A.E = E; A.kind = p->base_kernel; A.G = G; A.W = static_cast<double*>(Wb);
    if (p->base_kernel == GPSIG_BASE_RBF) hipLaunchKernelGGL(wide_tens_bwd_kernel<true>, dim3(unsigned(Tpad / 64), unsigned(Tpad < 65535 ? Tpad : 65535)), dim3(64), 0, c->stream, A);
    else hipLaunchKernelGGL(wide_tens_bwd_kernel<false>, dim3(unsigned(Tpad / 64), unsigned(Tpad < 65535 ? Tpad : 65535)), dim3(64), 0, c->stream, A);
    HIPCHK(c, hipGetLastError());
    // gZL_k (R, DA) = W_k ZR_k: column-major (DA x R) = ZR_k,cm (DA x R) W_k,cm (R x R);   gZR_k = W_k^T ZL_k: (DA x R) = ZL_k,cm W_k,cm^T
    CHK(dgemm_batched(c, false, false, DA, R, R, ZR, DA, R * DA, static_cast<const double*>(Wb), R, R * R, static_cast<double*>(gzl), DA, R * DA, lt));
    CHK(dgemm_batched(c, false, true, DA, R, R, ZL, DA, R * DA, static_cast<const double*>(Wb), R, R * R, static_cast<double*>(gzr), DA, R * DA, lt));
    const int64_t rows_out = int64_t(lt) * Tn * E;
    hipLaunchKernelGGL(wide_unaug_tens_kernel, dim3(grid_for(rows_out * d)), dim3(256), 0, c->stream, static_cast<const double*>(gzl), ZL,
                       static_cast<const double*>(gzr), ZR, rows_out, d, Tn, Tpad, E, gZ);
    HIPCHK(c, hipGetLastError());
    return GPSIG_OK;
}

}  // namespace gpsig
