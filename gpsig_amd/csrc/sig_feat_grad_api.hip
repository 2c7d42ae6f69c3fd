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
static __global__ void sig_sym_add_kernel(const double* __restrict__ G, int64_t N, double* __restrict__ S) {
    __shared__ double t[32][33];
    const int64_t i0 = int64_t(blockIdx.y) * 32, j0 = int64_t(blockIdx.x) * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int64_t i = j0 + r, j = i0 + threadIdx.x;                  // the transposed tile, read along ITS rows
        t[r][threadIdx.x] = (i < N && j < N) ? G[i * N + j] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int64_t i = i0 + r, j = j0 + threadIdx.x;
        if (i < N && j < N) S[i * N + j] = G[i * N + j] + t[threadIdx.x][r];
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
    ScaleParams s;
    memset(&s, 0, sizeof(s));
    s.d_in = d;                                   // the level primitives take their inputs as they come: no lengthscales, no lags
    auto features = [&](const double* Xs, int64_t N, int L, void* phi) -> int {
        SigFeatArgs A;
        memset(&A, 0, sizeof(A));
        A.X = Xs; A.N = N; A.L = L; A.difference = p->difference ? 1 : 0; A.P = s;
        A.w = nullptr; A.normalize = 0; A.jitter = 0.0; A.Phi = static_cast<double*>(phi); A.ld = ld; A.dlev = nullptr;
        A.order = order; A.natural_order = 1; A.unit_points = cosine ? 1 : 0;
        hipError_t e = ffn(A, unsigned(N < 4096 ? N : 4096), sig_features_lds_bytes(d, M, L), c->stream);
        if (e != hipSuccess) return fail(c, GPSIG_ERR_HIP, "sig_features_kernel: %s", hipGetErrorString(e));
        return GPSIG_OK;
    };
    CHK(features(X, N1, L1, phi1));
    if (two) CHK(features(Y, N2, L2, phi2));
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
    auto reverse = [&](const double* Xs, int64_t N, int L, const double* Ph, const double* dP, double* gx) -> int {
        SigFeatGradArgs A;
        memset(&A, 0, sizeof(A));
        A.X = Xs; A.N = N; A.L = L; A.difference = p->difference ? 1 : 0; A.Phi = Ph; A.dPhi = dP; A.ld = ld; A.gX = gx;
        A.unit_points = cosine ? 1 : 0;
        A.order = order;
        if (order > 1) {
            // E(dx) = sum_{k <= order} dx^(x)k / k!: its inverse series, and the weights of the Horner sub-steps' intermediates
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
        const size_t lds_rev = order > 1 ? sig_feat_grad_ho_lds_bytes(d, M, L) : sig_feat_grad_lds_bytes(d, M, L);
        hipError_t e = rfn(A, unsigned(N < 8192 ? N : 8192), lds_rev, c->stream);
        if (e != hipSuccess) return fail(c, GPSIG_ERR_HIP, "sig_feat_reverse_kernel: %s", hipGetErrorString(e));
        return GPSIG_OK;
    };
    CHK(reverse(X, N1, L1, P1, D1, gX));
    if (two) CHK(reverse(Y, N2, L2, P2, D2, gY));
    *done = true;
    return GPSIG_OK;
}

}  // namespace gpsig
