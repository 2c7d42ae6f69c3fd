// sig_feat_grad_api.hip -- the gradient of SignatureLinear's sequence-vs-sequence levels through the feature contraction (round 4).
//
// gpsig_seq_gram_levels_grad / gpsig_seq_diag_levels_grad (grad_api.hip) hand their device pointers to sig_features_grad below, which
// takes the call where the linear kernel's finite feature space makes it the cheaper evaluation:
//     Phi(X), Phi(Y)          sig_features_kernel, raw (no weights, no normalisation), natural order
//     dPhi_m(X) = G_m Phi_m(Y),  dPhi_m(Y) = G_m^T Phi_m(X)      one rocBLAS dgemm per level and side  (symmetric: (G_m + G_m^T) Phi_m(X), one;
//                                                                 diagonal: 2 G_m[i] Phi_m(x_i), elementwise)
//     gX, gY                  sig_feat_reverse_kernel / sig_feat_reverse_ho_kernel (sig_feat_grad_kernel.hpp): dPhi back through the feature sweep
//                             (first order / the higher orders' truncated-exponential steps; SignatureCosine: on the unit vectors)
// The pair kernels' reverse pass costs about three lattice sweeps per pair, L1 L2 (2d + 3M - 1) flops each; this costs 4 sum_m d^m per
// pair on the matrix cores plus a sweep per SEQUENCE.  Reference: TensorFlow's autodiff of signature_algs.py:8-35 behind
// kernels.py:208-237 (no gradient code of its own to cite).
#include "ctx.hpp"
#include "sig_feat_grad_kernel.hpp"

namespace gpsig {
typedef hipError_t (*SigFeatLaunchFn)(const SigFeatArgs&, unsigned, size_t, hipStream_t);
SigFeatLaunchFn sig_feat_lookup(int d, int M);                       // sig_feat_inst.hip
typedef hipError_t (*SigFeatGradLaunchFn)(const SigFeatGradArgs&, unsigned, size_t, hipStream_t);
SigFeatGradLaunchFn sig_feat_grad_pick_a(int d, int M);              // d = 1 .. 4     (sig_feat_grad_inst_a.hip)
SigFeatGradLaunchFn sig_feat_grad_pick_b(int d, int M);              // d = 5 .. 8
SigFeatGradLaunchFn sig_feat_grad_pick_c(int d, int M);              // d = 9 .. 12
SigFeatGradLaunchFn sig_feat_grad_pick_d(int d, int M);              // d = 13 .. 16
SigFeatGradLaunchFn sig_feat_grad_pick_e(int d, int M);              // d = 17 .. 24
SigFeatGradLaunchFn sig_feat_grad_pick_f(int d, int M);              // d = 25 .. 32
bool solver_dgemm(void** handle_slot, hipStream_t stream, bool transA, bool transB, int m, int n, int k, double alpha, const double* A, int lda,
                  const double* B, int ldb, double beta, double* C, int ldc, std::string* err);      // lowrank_solver.hip

static SigFeatGradLaunchFn sig_feat_grad_lookup(int d, int M) {
    if (d < 1 || d > 32) return nullptr;
    return d <= 4 ? sig_feat_grad_pick_a(d, M) : d <= 8 ? sig_feat_grad_pick_b(d, M) : d <= 12 ? sig_feat_grad_pick_c(d, M)
         : d <= 16 ? sig_feat_grad_pick_d(d, M) : d <= 24 ? sig_feat_grad_pick_e(d, M) : sig_feat_grad_pick_f(d, M);
}

// E(dx) = sum_{k <= order} dx^(x)k / k!: its inverse series, and the weights of the Horner sub-steps' intermediates
static void sig_ho_tables(int order, SigFeatGradArgs& A) {
    double fact[10];
    fact[0] = 1.0;
    for (int k = 1; k < 10; ++k) fact[k] = fact[k - 1] * k;
    A.cinv[0] = 1.0;
    for (int k = 1; k <= 8; ++k) {
        double v = 0.0;
        for (int j = 1; j <= k && j <= order; ++j) v -= A.cinv[k - j] / fact[j];
        A.cinv[k] = v;
    }
    for (int j = 0; j <= 8; ++j)
        for (int k = 0; k <= 8; ++k) A.w[j][k] = (j + k <= 9) ? fact[j] / fact[j + k] : 0.0;
}

// raw level features of N sequences (no weights, no normalisation), natural order, into Phi (N, ld); dlev: (N, M+1) level diagonals or NULL
static int sig_launch_features(gpsig_ctx* c, SigFeatLaunchFn ffn, const gpsig_params* p, int d, int order, bool cosine, int64_t ld, const double* Xs,
                               int64_t N, int L, double* phi, double* dlev) {
    const int M = p->num_levels;
    ScaleParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.d_in = d;                                  // the level primitives take their inputs as they come: no lengthscales, no lags
    SigFeatArgs A;
    memset(&A, 0, sizeof(A));
    A.X = Xs; A.N = N; A.L = L; A.difference = p->difference ? 1 : 0; A.P = sp;
    A.w = nullptr; A.normalize = 0; A.jitter = 0.0; A.Phi = phi; A.ld = ld; A.dlev = dlev;
    A.order = order; A.natural_order = 1; A.unit_points = cosine ? 1 : 0;
    const int64_t cap = sig_threads(d, M) <= 128 ? 16384 : 4096;      // (one- or two-wavefront workgroups are latency-bound at 16 per CU)
    hipError_t e = ffn(A, unsigned(N < cap ? N : cap), sig_features_lds_bytes(d, M, L), c->stream);
    if (e != hipSuccess) return fail(c, GPSIG_ERR_HIP, "sig_features_kernel: %s", hipGetErrorString(e));
    return GPSIG_OK;
}

// dPhi (N, ld) back through the feature sweep of N sequences: gx (N, L, d)
static int sig_launch_reverse(gpsig_ctx* c, SigFeatGradLaunchFn rfn, const gpsig_params* p, int d, int order, bool cosine, int64_t ld, const double* Xs,
                              int64_t N, int L, const double* Ph, const double* dP, double* gx) {
    const int M = p->num_levels;
    SigFeatGradArgs A;
    memset(&A, 0, sizeof(A));
    A.X = Xs; A.N = N; A.L = L; A.difference = p->difference ? 1 : 0; A.Phi = Ph; A.dPhi = dP; A.ld = ld; A.gX = gx;
    A.unit_points = cosine ? 1 : 0;
    A.order = order;
    if (order > 1) sig_ho_tables(order, A);
    const size_t lds_rev = order > 1 ? sig_feat_grad_ho_lds_bytes(d, M, L) : sig_feat_grad_lds_bytes(d, M, L);
    hipError_t e = rfn(A, unsigned(N < 8192 ? N : 8192), lds_rev, c->stream);
    if (e != hipSuccess) return fail(c, GPSIG_ERR_HIP, "sig_feat_reverse_kernel: %s", hipGetErrorString(e));
    return GPSIG_OK;
}

// dPhi[i][k] = 2 G[level(k)][i] Phi[i][k]: the diagonal K_m(x_i, x_i) = |Phi_m(x_i)|^2
static __global__ void sig_diag_dphi_kernel(const double* __restrict__ Phi, const double* __restrict__ G, int64_t N, int64_t ld, int D, int M,
                                            double* __restrict__ dPhi) {
    const int64_t total = N * ld;
    for (int64_t e = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t i = e / ld;
        const int k = int(e - i * ld);
        int m = 1, end = D, w = D;
        while (m <= M && k >= end) { ++m; w *= D; end += w; }
        dPhi[e] = m <= M ? 2.0 * G[int64_t(m) * N + i] * Phi[e] : 0.0;
    }
}

// S = G + G^T in 32 x 32 tiles through LDS (both reads along rows): the symmetric Gram's two roles of a sequence in one product
static __global__ void sig_sym_add_kernel(const double* __restrict__ G, int64_t N, double* __restrict__ S, int zero_diag = 0) {
    __shared__ double t[32][33];
    const int64_t i0 = int64_t(blockIdx.y) * 32, j0 = int64_t(blockIdx.x) * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int64_t i = j0 + r, j = i0 + threadIdx.x;                  // the transposed tile, read along ITS rows
        t[r][threadIdx.x] = (i < N && j < N) ? G[i * N + j] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int64_t i = i0 + r, j = j0 + threadIdx.x;
        if (i < N && j < N) S[i * N + j] = (zero_diag && i == j) ? 0.0 : G[i * N + j] + t[threadIdx.x][r];
    }
}

// ---- the level SUM's gradient (gpsig_kernel_K_grad below) ---------------------------------------------------------------------------
// K[i][j] = sum_m w_m <u_m(x_i), u_m(y_j)> with u_m = Phi_m / sqrt(|Phi_m|^2 + jitter) (normalised) or Phi_m: every level shares the
// upstream g, so ONE product P = g U(Y) over the whole feature width replaces the per-level products of the level primitive, and the
// (M+1, N1, N2) level arrays never exist.
struct SigLevelWeights { double w[9]; };

// U[i][k] = Phi[i][k] / sqrt(dlev[i][level(k)] + jitter)
static __global__ void sig_unit_levels_kernel(const double* __restrict__ Phi, const double* __restrict__ dlev, int64_t N, int64_t ld, int D, int M,
                                              double jitter, double* __restrict__ U) {
    __shared__ double inv[9];
    for (int64_t i = blockIdx.x; i < N; i += gridDim.x) {
        __syncthreads();
        if (threadIdx.x <= unsigned(M)) inv[threadIdx.x] = threadIdx.x == 0 ? 0.0 : 1.0 / sqrt(dlev[i * (M + 1) + threadIdx.x] + jitter);
        __syncthreads();
        const double* ph = Phi + i * ld;
        double* u = U + i * ld;
        int off = 0, w = D;
        for (int m = 1; m <= M; ++m) {
            const double sc = inv[m];
            for (int k = threadIdx.x; k < w; k += blockDim.x) u[off + k] = sc * ph[off + k];
            off += w;
            w *= D;
        }
    }
}

// One workgroup per row i.  c_m = <U_m[i], P_m[i]> goes to cdot (sum_j g[i][j] Kn_m[i][j]: the weights' gradient, and the part of the
// upstream that the normalisation absorbs);  P_m[i] <- w_m (P_m[i] - U_m[i] c_m) / s_m[i]  (normalised; u = Phi / s, s^2 = |Phi|^2 + jitter)
// or w_m P_m[i]: the gradient with respect to the raw features, where the reverse sweep starts.
static __global__ void sig_sum_grad_rows_kernel(const double* __restrict__ U, double* __restrict__ P, const double* __restrict__ dlev, int64_t N,
                                                int64_t ld, int D, int M, double jitter, int normalize, SigLevelWeights W, double* __restrict__ cdot) {
    __shared__ double red[16];
    __shared__ double cm;
    for (int64_t i = blockIdx.x; i < N; i += gridDim.x) {
        const double* u = U + i * ld;
        double* pr = P + i * ld;
        int off = 0, w = D;
        for (int m = 1; m <= M; ++m) {
            double s = 0.0;
            for (int k = threadIdx.x; k < w; k += blockDim.x) s = fma(u[off + k], pr[off + k], s);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            __syncthreads();                                   // (red / cm of the level before are no longer read)
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0.0;
                for (unsigned q = 0; q < (blockDim.x + 63) / 64; ++q) t += red[q];
                cm = t;
                if (cdot) cdot[i * (M + 1) + m] = t;
            }
            __syncthreads();
            const double c = cm;
            if (normalize) {
                const double sc = W.w[m] / sqrt(dlev[i * (M + 1) + m] + jitter);
                for (int k = threadIdx.x; k < w; k += blockDim.x) pr[off + k] = sc * (pr[off + k] - u[off + k] * c);
            } else {
                const double sc = W.w[m];
                for (int k = threadIdx.x; k < w; k += blockDim.x) pr[off + k] *= sc;
            }
            off += w;
            w *= D;
        }
        if (cdot && threadIdx.x == 0) cdot[i * (M + 1)] = 0.0;
    }
}

// gw[m] = scale * sum_i cdot[i][m] in a fixed order (one workgroup)
static __global__ void sig_weight_grad_kernel(const double* __restrict__ cdot, int64_t N, int M, double scale, double* __restrict__ gw) {
    __shared__ double red[256];
    for (int m = 0; m <= M; ++m) {
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < N; i += 256) s += cdot[i * (M + 1) + m];
        __syncthreads();
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (int(threadIdx.x) < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) gw[m] = scale * red[0];
    }
}

// *done = false leaves the call to the pair kernels.  X, Y, G, gX, gY: device pointers; G (M+1, N1, N2) row-major, or (M+1, N1) when diag.
int sig_features_grad(gpsig_ctx* c, const gpsig_params* p, int d, const double* X, const double* Y, int64_t N1, int64_t N2, int L1, int L2, bool diag,
                      bool sym, const double* G, double* gX, double* gY, bool* done) {
    *done = false;
    const int M = p->num_levels;
    // the linear kernel, and the cosine kernel as the linear kernel of the unit vectors x / |x| (kernels.py:820-828)
    const bool cosine = p->base_kernel == GPSIG_BASE_COSINE;
    const int order = p->order < 1 ? 1 : (p->order > M ? M : p->order);
    if (c->sig_features_grad == 0 || !(p->base_kernel == GPSIG_BASE_LINEAR || cosine) || M < 2 || M > 8 || c->capturing) return GPSIG_OK;
    if (N1 <= 0 || N2 <= 0 || N1 > 0x3fffffff || N2 > 0x3fffffff) return GPSIG_OK;
    SigFeatLaunchFn ffn = sig_feat_lookup(d, M);
    SigFeatGradLaunchFn rfn = sig_feat_grad_lookup(d, M);
    if (!ffn || !rfn) return GPSIG_OK;
    const int r1 = p->difference ? L1 - 1 : L1, r2 = p->difference ? L2 - 1 : L2;
    if (r1 < 1 || r2 < 1) return GPSIG_OK;
    const int64_t F = sig_feature_count(d, M), ld = (F + 1 + 15) / 16 * 16;
    const int Lmax = L1 > L2 ? L1 : L2;
    const size_t lds_f = sig_features_lds_bytes(d, M, Lmax);
    const size_t lds_r = order > 1 ? sig_feat_grad_ho_lds_bytes(d, M, Lmax) : sig_feat_grad_lds_bytes(d, M, Lmax);
    if (lds_f > 150 * 1024 || lds_r > 158 * 1024) return GPSIG_OK;
    const bool two = !diag && !sym;
    const size_t bytes = sizeof(double) * size_t(ld) * (size_t(N1) + (two ? size_t(N2) : 0)) * 2 + (sym ? sizeof(double) * size_t(N1) * N1 : 0);
    if (bytes > (size_t(48) << 30)) return GPSIG_OK;
    if (c->sig_features_grad < 0) {
        // the pair kernels' reverse pass: about three lattice sweeps per pair at a third of the vector peak; this route: two products of depth
        // F per pair near the matrix peak (one with the symmetrised upstream; the diagonal: none) and, per sequence, the forward sweep plus a reverse sweep of three times its
        // arithmetic at a fraction of the vector rate; a dozen launches against one or two
        const double pairs = diag ? double(N1) : double(N1) * double(N2), seqs = double(N1) + (two ? double(N2) : 0.0);
        // (cosine: the point kernels' two-step reverse pass; higher orders: the scratch-based kernels, hundreds of launches per block of pairs)
        const double lattice = 3.0 * double(r1) * r2 * (2.0 * d + 3.0 * M - 1.0) * (cosine ? 1.5 : 1.0) * (order > 1 ? 50.0 : 1.0);
        const double t_lat = 100e-6 + pairs * lattice / 20e12;
        const double t_feat = 150e-6 + 15e-6 * M + (diag ? 0.0 : pairs * (sym ? 2.0 : 4.0) * double(F) / 45e12) + seqs * double(Lmax) * double(sig_ipow(d, M)) * 8.0 / 8e12;
        if (!(t_feat < t_lat)) return GPSIG_OK;
    }
    void *phi1, *phi2 = nullptr, *dphi1, *dphi2 = nullptr;
    CHK(ensure(c, B_SF0, sizeof(double) * size_t(ld) * N1 + 64, &phi1));
    CHK(ensure(c, B_SF3, sizeof(double) * size_t(ld) * N1 + 64, &dphi1));
    if (two) {
        CHK(ensure(c, B_SF1, sizeof(double) * size_t(ld) * N2 + 64, &phi2));
        CHK(ensure(c, B_SF4, sizeof(double) * size_t(ld) * N2 + 64, &dphi2));
    }
    void* gsym = nullptr;
    if (sym) CHK(ensure(c, B_SF5, sizeof(double) * size_t(N1) * N1 + 64, &gsym));
    c->sf_valid = false;                          // (B_SF0 no longer holds what "sig_features_keep" remembers)
    CHK(sig_launch_features(c, ffn, p, d, order, cosine, ld, X, N1, L1, static_cast<double*>(phi1), nullptr));
    if (two) CHK(sig_launch_features(c, ffn, p, d, order, cosine, ld, Y, N2, L2, static_cast<double*>(phi2), nullptr));
    const double* P1 = static_cast<const double*>(phi1);
    const double* P2 = two ? static_cast<const double*>(phi2) : P1;
    double* D1 = static_cast<double*>(dphi1);
    double* D2 = static_cast<double*>(dphi2);
    if (diag) {
        hipLaunchKernelGGL(sig_diag_dphi_kernel, dim3(grid_for(N1 * ld)), dim3(256), 0, c->stream, P1, G, N1, ld, d, M, D1);
        HIPCHK(c, hipGetLastError());
    } else {
        // the columns behind the levels (level 0, padding) carry no gradient and are never read by the reverse kernel
        std::string err;
        int off = 0, w = d;
        for (int m = 1; m <= M; ++m) {
            const double* Gm = G + int64_t(m) * N1 * N2;
            if (sym) {      // k(x_i, x_j) = k(x_j, x_i): x_i collects G[i][j] + G[j][i] -- one product with the symmetrised upstream instead of two
                hipLaunchKernelGGL(sig_sym_add_kernel, dim3(unsigned((N1 + 31) / 32), unsigned((N1 + 31) / 32)), dim3(32, 8), 0, c->stream, Gm, N1,
                                   static_cast<double*>(gsym));
                HIPCHK(c, hipGetLastError());
                Gm = static_cast<const double*>(gsym);
            }
            // row-major C (N1 x w) = G_m (N1 x N2) B (N2 x w)  ==  column-major C^T (w x N1) = B^T (w x N2, ld) G_m^T (N2 x N1, ld N2)
            if (!solver_dgemm(&c->blas_handle, c->stream, false, false, w, int(N1), int(N2), 1.0, P2 + off, int(ld), Gm, int(N2), 0.0, D1 + off, int(ld), &err))
                return fail(c, GPSIG_ERR_HIP, "%s", err.c_str());
            // row-major C' (N2 x w) = G_m^T B1 (N1 x w)  ==  column-major C'^T (w x N2) = B1^T (w x N1, ld) G_m (the stored N2 x N1 matrix transposed)
            if (!sym) {
                if (!solver_dgemm(&c->blas_handle, c->stream, false, true, w, int(N2), int(N1), 1.0, P1 + off, int(ld), Gm, int(N2), 0.0, D2 + off, int(ld), &err))
                    return fail(c, GPSIG_ERR_HIP, "%s", err.c_str());
            }
            off += w;
            w *= d;
        }
    }
    CHK(sig_launch_reverse(c, rfn, p, d, order, cosine, ld, X, N1, L1, P1, D1, gX));
    if (two) CHK(sig_launch_reverse(c, rfn, p, d, order, cosine, ld, Y, N2, L2, P2, D2, gY));
    *done = true;
    return GPSIG_OK;
}

// The gradient of gpsig_kernel_K's level sum through the feature space: *done = false leaves the call to the caller's fallback (the level
// primitives).  g (N1, N2) upstream; gw (M+1, device) receives d/d(sigma variances[m]) for m >= 1 from the off-diagonal entries (level 0
// and a normalised symmetric Gram's constant diagonal are the caller's: sums of g); probe: only say whether the route applies.
int sig_features_sum_grad(gpsig_ctx* c, const gpsig_params* p, int d, const double* X, const double* Y, int64_t N1, int64_t N2, int L1, int L2,
                          const double* g, double* gX, double* gY, double* gw, bool probe, bool* done) {
    *done = false;
    const int M = p->num_levels;
    const bool cosine = p->base_kernel == GPSIG_BASE_COSINE;
    const bool sym = Y == nullptr;
    const int order = p->order < 1 ? 1 : (p->order > M ? M : p->order);
    if (c->sig_features_grad == 0 || !(p->base_kernel == GPSIG_BASE_LINEAR || cosine) || M < 2 || M > 8 || c->capturing) return GPSIG_OK;
    if (N1 <= 0 || N2 <= 0 || N1 > 0x3fffffff || N2 > 0x3fffffff) return GPSIG_OK;
    SigFeatLaunchFn ffn = sig_feat_lookup(d, M);
    SigFeatGradLaunchFn rfn = sig_feat_grad_lookup(d, M);
    if (!ffn || !rfn) return GPSIG_OK;
    const int r1 = p->difference ? L1 - 1 : L1, r2 = p->difference ? L2 - 1 : L2;
    if (r1 < 1 || r2 < 1) return GPSIG_OK;
    const int64_t F = sig_feature_count(d, M), ld = (F + 1 + 15) / 16 * 16;
    const int Lmax = L1 > L2 ? L1 : L2;
    const size_t lds_f = sig_features_lds_bytes(d, M, Lmax);
    const size_t lds_r = order > 1 ? sig_feat_grad_ho_lds_bytes(d, M, Lmax) : sig_feat_grad_lds_bytes(d, M, Lmax);
    if (lds_f > 150 * 1024 || lds_r > 158 * 1024) return GPSIG_OK;
    const bool norm = p->normalization != 0;
    const size_t rows = size_t(N1) + (sym ? 0 : size_t(N2));
    const size_t bytes = sizeof(double) * size_t(ld) * rows * (norm ? 3 : 2) + (sym ? sizeof(double) * size_t(N1) * N1 : 0);
    if (bytes > (size_t(48) << 30)) return GPSIG_OK;
    if (c->sig_features_grad < 0) {       // the same comparison as the level primitive's, with ONE product per side (see there)
        const double pairs = double(N1) * double(N2), seqs = double(rows);
        const double lattice = 3.0 * double(r1) * r2 * (2.0 * d + 3.0 * M - 1.0) * (cosine ? 1.5 : 1.0) * (order > 1 ? 50.0 : 1.0);
        const double t_lat = 100e-6 + pairs * lattice / 20e12;
        const double t_feat = 120e-6 + pairs * (sym ? 2.0 : 4.0) * double(F) / 45e12 + seqs * double(Lmax) * double(sig_ipow(d, M)) * 8.0 / 8e12;
        if (!(t_feat < t_lat)) return GPSIG_OK;
    }
    if (probe) { *done = true; return GPSIG_OK; }
    void *phi1, *phi2 = nullptr, *dphi1, *dphi2 = nullptr, *u1 = nullptr, *u2 = nullptr, *small;
    CHK(ensure(c, B_SF0, sizeof(double) * size_t(ld) * N1 + 64, &phi1));
    CHK(ensure(c, B_SF3, sizeof(double) * size_t(ld) * N1 + 64, &dphi1));
    if (norm) CHK(ensure(c, B_GR0, sizeof(double) * size_t(ld) * N1 + 64, &u1));
    if (!sym) {
        CHK(ensure(c, B_SF1, sizeof(double) * size_t(ld) * N2 + 64, &phi2));
        CHK(ensure(c, B_SF4, sizeof(double) * size_t(ld) * N2 + 64, &dphi2));
        if (norm) CHK(ensure(c, B_GR1, sizeof(double) * size_t(ld) * N2 + 64, &u2));
    }
    CHK(ensure(c, B_GR2, sizeof(double) * (M + 1) * (2 * size_t(N1) + size_t(N2)) + 64, &small));
    double* dl1 = static_cast<double*>(small);                   // raw level diagonals of X, of Y, the rows' dot products
    double* dl2 = dl1 + size_t(N1) * (M + 1);
    double* cdot = dl2 + size_t(N2) * (M + 1);
    void* gsym = nullptr;
    if (sym) CHK(ensure(c, B_SF5, sizeof(double) * size_t(N1) * N1 + 64, &gsym));
    c->sf_valid = false;
    CHK(sig_launch_features(c, ffn, p, d, order, cosine, ld, X, N1, L1, static_cast<double*>(phi1), dl1));
    if (!sym) CHK(sig_launch_features(c, ffn, p, d, order, cosine, ld, Y, N2, L2, static_cast<double*>(phi2), dl2));
    const double* P1 = static_cast<const double*>(phi1);
    const double* P2 = sym ? P1 : static_cast<const double*>(phi2);
    const double *U1 = P1, *U2 = P2;
    if (norm) {
        hipLaunchKernelGGL(sig_unit_levels_kernel, dim3(unsigned(N1 < 8192 ? N1 : 8192)), dim3(256), 0, c->stream, P1, dl1, N1, ld, d, M, p->jitter,
                           static_cast<double*>(u1));
        HIPCHK(c, hipGetLastError());
        U1 = static_cast<const double*>(u1);
        U2 = U1;
        if (!sym) {
            hipLaunchKernelGGL(sig_unit_levels_kernel, dim3(unsigned(N2 < 8192 ? N2 : 8192)), dim3(256), 0, c->stream, P2, dl2, N2, ld, d, M, p->jitter,
                               static_cast<double*>(u2));
            HIPCHK(c, hipGetLastError());
            U2 = static_cast<const double*>(u2);
        }
    }
    double* D1 = static_cast<double*>(dphi1);
    double* D2 = static_cast<double*>(dphi2);
    const double* S = g;
    if (sym) {      // both roles of a sequence in one product; a normalised Gram's diagonal is the constant sum of the weights
        hipLaunchKernelGGL(sig_sym_add_kernel, dim3(unsigned((N1 + 31) / 32), unsigned((N1 + 31) / 32)), dim3(32, 8), 0, c->stream, g, N1,
                           static_cast<double*>(gsym), norm ? 1 : 0);
        HIPCHK(c, hipGetLastError());
        S = static_cast<const double*>(gsym);
    }
    std::string err;
    // row-major P (N1 x F) = S (N1 x N2) U2 (N2 x F)  ==  column-major P^T (F x N1) = U2^T (F x N2, ld) S^T (N2 x N1, ld N2)
    if (!solver_dgemm(&c->blas_handle, c->stream, false, false, int(F), int(N1), int(N2), 1.0, U2, int(ld), S, int(N2), 0.0, D1, int(ld), &err))
        return fail(c, GPSIG_ERR_HIP, "%s", err.c_str());
    if (!sym && !solver_dgemm(&c->blas_handle, c->stream, false, true, int(F), int(N2), int(N1), 1.0, U1, int(ld), S, int(N2), 0.0, D2, int(ld), &err))
        return fail(c, GPSIG_ERR_HIP, "%s", err.c_str());
    SigLevelWeights W;
    for (int m = 0; m <= 8; ++m) W.w[m] = m <= M ? p->sigma * p->variances[m] : 0.0;
    hipLaunchKernelGGL(sig_sum_grad_rows_kernel, dim3(unsigned(N1 < 8192 ? N1 : 8192)), dim3(256), 0, c->stream, U1, D1, dl1, N1, ld, d, M, p->jitter,
                       norm ? 1 : 0, W, cdot);
    HIPCHK(c, hipGetLastError());
    if (!sym) {
        hipLaunchKernelGGL(sig_sum_grad_rows_kernel, dim3(unsigned(N2 < 8192 ? N2 : 8192)), dim3(256), 0, c->stream, U2, D2, dl2, N2, ld, d, M, p->jitter,
                           norm ? 1 : 0, W, static_cast<double*>(nullptr));
        HIPCHK(c, hipGetLastError());
    }
    if (gw) {
        hipLaunchKernelGGL(sig_weight_grad_kernel, dim3(1), dim3(256), 0, c->stream, cdot, N1, M, sym ? 0.5 : 1.0, gw);
        HIPCHK(c, hipGetLastError());
    }
    CHK(sig_launch_reverse(c, rfn, p, d, order, cosine, ld, X, N1, L1, P1, D1, gX));
    if (!sym) CHK(sig_launch_reverse(c, rfn, p, d, order, cosine, ld, Y, N2, L2, P2, D2, gY));
    *done = true;
    return GPSIG_OK;
}

}  // namespace gpsig

extern "C" int gpsig_kernel_K_grad(gpsig_ctx* c, const gpsig_params* p, const void* X, const void* X2, int64_t N1, int64_t N2, int32_t L1, int32_t L2,
                                   const void* g, void* gX, void* gX2, void* g_weights, int32_t* taken) {
    using namespace gpsig;
    if (!c || !p || !taken) return GPSIG_ERR_INVALID;
    *taken = 0;
    if (p->dtype != GPSIG_F64) return fail(c, GPSIG_ERR_INVALID, "gpsig_kernel_K_grad: float64 only");
    if (c->ptr_mode != GPSIG_PTR_DEVICE) return fail(c, GPSIG_ERR_INVALID, "gpsig_kernel_K_grad takes device pointers");
    if (p->lengthscales || p->num_lags != 0) return fail(c, GPSIG_ERR_INVALID, "gpsig_kernel_K_grad takes its inputs as they come (no lengthscales, no lags), like the level primitives");
    if (!p->variances || p->num_levels < 1 || p->num_features < 1) return fail(c, GPSIG_ERR_INVALID, "gpsig_kernel_K_grad: bad parameters");
    const bool probe = g == nullptr;
    if (!X || (!probe && (!gX || (X2 && !gX2)))) return fail(c, GPSIG_ERR_INVALID, "gpsig_kernel_K_grad: null argument");
    HIPCHK(c, hipSetDevice(c->device));
    bool done = false;
    CHK(sig_features_sum_grad(c, p, p->num_features, static_cast<const double*>(X), static_cast<const double*>(X2), N1, X2 ? N2 : N1, L1,
                              X2 ? L2 : L1, static_cast<const double*>(g), static_cast<double*>(gX), static_cast<double*>(gX2),
                              static_cast<double*>(g_weights), probe, &done));
    *taken = done ? 1 : 0;
    return GPSIG_OK;
}

// ---- the explicit level features as an op of their own (round 4) ----------------------------------------------------------------------
// Phi(x) = (Phi_1, .., Phi_M)(x), Phi_m in (R^d)^(x)m: for SignatureLinear K_m(x, y) = <Phi_m(x), Phi_m(y)> (signature_algs.py:8-35 unrolled;
// order > 1: the truncated-exponential steps of :37-74, order = num_levels and difference on: the signature of the piecewise-linear path, what
// the reference's notebook checks against esig).  With them <z_1 (x) .. (x) z_m, Phi_m(x)> is the tensor-vs-sequence kernel of a rank-one
// inducing tensor (signature_algs.py:101-127), so the linear kernel's Kzx is a plain product of (T, F) and (N, F) matrices.
namespace gpsig {
static bool sig_features_plan(const gpsig_params* p, int L, SigFeatLaunchFn* ffn, SigFeatGradLaunchFn* rfn, int64_t* ld, int* order, bool* cosine) {
    const int M = p->num_levels, d = p->num_features;
    *cosine = p->base_kernel == GPSIG_BASE_COSINE;
    *order = p->order < 1 ? 1 : (p->order > M ? M : p->order);
    if (!(p->base_kernel == GPSIG_BASE_LINEAR || *cosine) || M < 2 || M > 8 || d < 1 || d > 32) return false;
    if (p->lengthscales || p->num_lags != 0 || p->dtype != GPSIG_F64) return false;
    *ffn = sig_feat_lookup(d, M);
    *rfn = sig_feat_grad_lookup(d, M);
    if (!*ffn || !*rfn) return false;
    if ((p->difference ? L - 1 : L) < 1) return false;
    const size_t lds_f = sig_features_lds_bytes(d, M, L);
    const size_t lds_r = *order > 1 ? sig_feat_grad_ho_lds_bytes(d, M, L) : sig_feat_grad_lds_bytes(d, M, L);
    if (lds_f > 150 * 1024 || lds_r > 158 * 1024) return false;
    *ld = (int64_t(sig_feature_count(d, M)) + 1 + 15) / 16 * 16;
    return true;
}
}  // namespace gpsig

extern "C" int64_t gpsig_seq_features_ld(const gpsig_params* p, int32_t L) {
    using namespace gpsig;
    SigFeatLaunchFn ffn; SigFeatGradLaunchFn rfn; int64_t ld = 0; int order; bool cosine;
    if (!p || !sig_features_plan(p, L, &ffn, &rfn, &ld, &order, &cosine)) return 0;
    return ld;
}

extern "C" int gpsig_seq_features(gpsig_ctx* c, const gpsig_params* p, const void* X, int64_t N, int32_t L, void* out) {
    using namespace gpsig;
    if (!c || !p) return GPSIG_ERR_INVALID;
    SigFeatLaunchFn ffn; SigFeatGradLaunchFn rfn; int64_t ld = 0; int order; bool cosine;
    if (!sig_features_plan(p, L, &ffn, &rfn, &ld, &order, &cosine))
        return fail(c, GPSIG_ERR_UNSUPPORTED, "gpsig_seq_features: linear / cosine kernel, float64, 2 <= num_levels <= 8, d <= 32, inputs as they come, a sequence's arrays within the LDS (gpsig_seq_features_ld says 0 otherwise)");
    if (N <= 0 || !X || !out) return fail(c, GPSIG_ERR_INVALID, "gpsig_seq_features: null or empty argument");
    HIPCHK(c, hipSetDevice(c->device));
    const int d = p->num_features;
    const void* dx;
    void* dout;
    CHK(in_dev(c, B_IN0, X, sizeof(double) * size_t(N) * L * d, &dx));
    CHK(out_dev(c, B_OUT0, out, sizeof(double) * size_t(N) * ld, &dout));
    CHK(sig_launch_features(c, ffn, p, d, order, cosine, ld, static_cast<const double*>(dx), N, L, static_cast<double*>(dout), nullptr));
    CHK(out_done(c, out, dout, sizeof(double) * size_t(N) * ld));
    return finish(c);
}

extern "C" int gpsig_seq_features_grad(gpsig_ctx* c, const gpsig_params* p, const void* X, int64_t N, int32_t L, const void* Phi, const void* dPhi,
                                       void* gX) {
    using namespace gpsig;
    if (!c || !p) return GPSIG_ERR_INVALID;
    SigFeatLaunchFn ffn; SigFeatGradLaunchFn rfn; int64_t ld = 0; int order; bool cosine;
    if (!sig_features_plan(p, L, &ffn, &rfn, &ld, &order, &cosine)) return fail(c, GPSIG_ERR_UNSUPPORTED, "gpsig_seq_features_grad: see gpsig_seq_features");
    if (c->ptr_mode != GPSIG_PTR_DEVICE) return fail(c, GPSIG_ERR_INVALID, "gpsig_seq_features_grad takes device pointers");
    if (N <= 0 || !X || !Phi || !dPhi || !gX) return fail(c, GPSIG_ERR_INVALID, "gpsig_seq_features_grad: null or empty argument");
    HIPCHK(c, hipSetDevice(c->device));
    return sig_launch_reverse(c, rfn, p, p->num_features, order, cosine, ld, static_cast<const double*>(X), N, L, static_cast<const double*>(Phi),
                              static_cast<const double*>(dPhi), static_cast<double*>(gX));
}
