// tvs_grad_api.hip -- host side of the tile kernel for the reverse pass of the tensor-vs-sequence chains (tvs_grad_tile_kernel.hpp):
// preparation of the operands in the forward tile kernel's layouts, launch, reduction of the partial sums.  Called by
// gpsig_tens_vs_seq_levels_grad (grad_api.hip) with device pointers; *done = false leaves the call to the older kernels.
#include "ctx.hpp"
#include "tvs_tile_kernel.hpp"
#include "tvs_grad_tile_kernel.hpp"

namespace gpsig {
typedef hipError_t (*TvsGradTileLaunchFn)(const TvsGradTileArgs&, dim3, size_t, hipStream_t);
TvsGradTileLaunchFn tvs_grad_tile_lookup_m1(int, int, bool);
TvsGradTileLaunchFn tvs_grad_tile_lookup_m2(int, int, bool);
TvsGradTileLaunchFn tvs_grad_tile_lookup_m3(int, int, bool);
TvsGradTileLaunchFn tvs_grad_tile_lookup_m4(int, int, bool);
TvsGradTileLaunchFn tvs_grad_tile_lookup_m5(int, int, bool);
TvsGradTileLaunchFn tvs_grad_tile_lookup_m6(int, int, bool);
TvsGradTileLaunchFn tvs_grad_tile_lookup_ho(int M, int D, bool paired, int kind);      // tvs_grad_tile_inst_ho.hip: SignatureRBF and the Matern families, order > 1
int tvs_tile_width(int d);

static TvsGradTileLaunchFn tvs_grad_tile_lookup(int M, int D, int kind, bool paired) {
    switch (M) {
        case 1: return tvs_grad_tile_lookup_m1(D, kind, paired);
        case 2: return tvs_grad_tile_lookup_m2(D, kind, paired);
        case 3: return tvs_grad_tile_lookup_m3(D, kind, paired);
        case 4: return tvs_grad_tile_lookup_m4(D, kind, paired);
        case 5: return tvs_grad_tile_lookup_m5(D, kind, paired);
        case 6: return tvs_grad_tile_lookup_m6(D, kind, paired);
        default: return nullptr;
    }
}

static __global__ void tvs_grad_sum_kernel(const double* __restrict__ part, int64_t n, double* __restrict__ out) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 64) s += part[i];
    s = grad_wave_sum(s);
    if (threadIdx.x == 0) out[0] += s;
}

// does the tile kernel hold a higher-order instance for this call?  (grad_api.hip prefers it to the wide route's chains at these widths)
bool tvs_grad_tile_ho_available(const gpsig_ctx* c, const gpsig_params* p, int d, int L, int increments) {
    const int M = p->num_levels;
    const bool mat = c->tvs_grad_matern != 0 && (p->base_kernel == GPSIG_BASE_MATERN12 || p->base_kernel == GPSIG_BASE_MATERN32 || p->base_kernel == GPSIG_BASE_MATERN52);
    if (!(p->order > 1 && M > 1) || !(p->base_kernel == GPSIG_BASE_RBF || mat) || (p->order < M ? p->order : M) > TVSG_MAX_ORDER) return false;
    if (c->grad_impl != 0 || c->tvs_grad_tile == 0) return false;
    const int D = tvs_tile_width(d);
    if (D == 0 || !tvs_grad_tile_lookup_ho(M, D, increments != 0, mat ? TVSG_MATERN : BASE_RBF)) return false;
    const int rec_elems = (L * D + L + TVSG_REC_ALIGN - 1) / TVSG_REC_ALIGN * TVSG_REC_ALIGN;
    return tvs_grad_tile_lds_bytes(D, rec_elems, true) <= 64 * 1024;
}

// Z (lt, T, E_in, d), X (N, L, d) on the device, already scaled; gZ, gX overwritten; gb: 2 doubles on the device ([0] accumulated) or NULL.
// fac == NULL: G (M+1, T, N) is the upstream gradient of the level array.  fac (N, M+1): G (T, N) is the upstream gradient of the
// weighted level sum  sum_m fac[n][m] level_m[t][n],  and gfac (N, M+1) receives the gradient with respect to the factors.
// aux (N, lt, Tpad) or NULL: the chain totals the forward tile kernel left (tvs_tile_kernel.hpp), which save the forward sweep here.
int tvs_grad_tile_device(gpsig_ctx* c, const gpsig_params* p, int d, const double* Z, const double* X, const double* G, int64_t Tn, int64_t N,
                         int L, int increments, const double* fac, const double* aux, double* gZ, double* gX, double* gfac, double* gb, size_t budget, bool* done) {
    *done = false;
    const int M = p->num_levels, lt = M * (M + 1) / 2;
    const bool ho = p->order > 1 && M > 1;
    const bool is_mat = p->base_kernel == GPSIG_BASE_MATERN12 || p->base_kernel == GPSIG_BASE_MATERN32 || p->base_kernel == GPSIG_BASE_MATERN52;
    if (ho && (!(p->base_kernel == GPSIG_BASE_RBF || (is_mat && c->tvs_grad_matern != 0)) || (p->order < M ? p->order : M) > TVSG_MAX_ORDER)) return GPSIG_OK;
    const int D = tvs_tile_width(d);
    if (D == 0 || M > 6 || Tn < 1 || N < 1) return GPSIG_OK;
    const bool matern = c->tvs_grad_matern != 0 && is_mat;
    const int kind = p->base_kernel == GPSIG_BASE_LINEAR ? BASE_LINEAR : (p->base_kernel == GPSIG_BASE_RBF ? BASE_RBF : (matern ? TVSG_MATERN : -1));
    const bool collapse = increments && kind == BASE_LINEAR;              // <x, z1> - <x, z0> = <x, z1 - z0>
    const bool paired = increments && !collapse;
    const int E = paired ? 2 : 1;
    TvsGradTileLaunchFn fn = ho ? tvs_grad_tile_lookup_ho(M, D, paired, kind) : tvs_grad_tile_lookup(M, D, kind, paired);
    if (!fn) return GPSIG_OK;
    const int NR = tvs_grad_tile_roles(M, kind);
    const int rec_elems = (L * D + L + TVSG_REC_ALIGN - 1) / TVSG_REC_ALIGN * TVSG_REC_ALIGN;
    const size_t lds = tvs_grad_tile_lds_bytes(D, rec_elems, tvsg_has_table(kind));
    if (lds > 64 * 1024) return GPSIG_OK;
    const int64_t Tpad = (Tn + 63) / 64 * 64;
    const int tpw = paired ? 32 : 64;
    const int64_t TB = (Tn + tpw - 1) / tpw;
    if (TB > 65535) return GPSIG_OK;
    // sequences per workgroup: enough workgroups to fill the chip a few times over, runs long enough to amortise a lane's load of its components
    int64_t runs = 4096 / TB < 1 ? 1 : 4096 / TB;
    if (runs > N) runs = N;
    int64_t run = (N + runs - 1) / runs;
    // (runs of at least 8 amortise a lane's load of its components -- unless that leaves most of the chip idle: a minibatch of 50 sequences against
    // 500 tensors with increments is 16 x 7 x 3 = 336 wavefronts of 8 x 2 sweeps over the sequence, 16 x 50 x 3 = 2,400 of one)
    if (run < 8 && N >= 8) {
        run = 8;
        while (run > 1 && TB * NR * ((N + run - 1) / run) < 2048) run /= 2;
    }
    if (run > 64) run = 64;
    runs = (N + run - 1) / run;
    // the d/dx partial sums of all tensor blocks are bounded by the scratch budget: sequences in chunks of whole runs
    const size_t gx_per_seq = sizeof(double) * size_t(NR) * TB * L * D;
    int64_t chunk_runs = int64_t(budget / (gx_per_seq * size_t(run) ? gx_per_seq * size_t(run) : 1));
    if (chunk_runs < 1) chunk_runs = 1;
    if (chunk_runs > runs) chunk_runs = runs;
    if (chunk_runs > 65535) chunk_runs = 65535;
    const double mat_c = p->base_kernel == GPSIG_BASE_MATERN12 ? 1.0 : (p->base_kernel == GPSIG_BASE_MATERN32 ? 1.7320508075688772935 : 2.2360679774997896964);
    const double pre = kind == BASE_RBF ? EXP_PRESCALE256 : (kind == TVSG_MATERN ? mat_c * 256.0 / 0x1.62e42fefa39efp-1 : 1.0);
    ScaleParams s;
    memset(&s, 0, sizeof(s));
    s.d_in = d;
    const int GL = fac ? 1 : M + 1;                                           // level slots of the upstream gradient
    void *zl, *zn, *xr, *gt, *gxp, *gzp, *gbp = nullptr, *gfp = nullptr;
    CHK(ensure(c, B_GR0, sizeof(double) * size_t(lt) * E * D * Tpad + 8, &zl));
    CHK(ensure(c, B_GR1, sizeof(double) * size_t(lt) * E * Tpad + 8, &zn));
    CHK(ensure(c, B_GR2, sizeof(double) * size_t(N) * rec_elems + 8, &xr));
    CHK(ensure(c, B_GR3, sizeof(double) * size_t(N) * GL * Tpad + 8, &gt));
    CHK(ensure(c, B_GR4, gx_per_seq * size_t(chunk_runs) * size_t(run) + 8, &gxp));
    const size_t gz_stride = size_t(lt) * E * D * Tpad;
    CHK(ensure(c, B_GR5, sizeof(double) * gz_stride * size_t(runs) + 8, &gzp));
    if (gb) CHK(ensure(c, B_GR6, sizeof(double) * size_t(TB) * runs * NR + 8, &gbp));
    if (fac) CHK(ensure(c, B_GR7B, sizeof(double) * size_t(TB) * N * (M + 1) + 8, &gfp));
    hipLaunchKernelGGL(prep_tensors_tile_kernel, dim3(grid_for(Tpad * lt * E)), dim3(256), 0, c->stream, Z, lt, Tn, Tpad, increments ? 2 : 1,
                       collapse ? 1 : 0, pre, s, D, static_cast<double*>(zl), static_cast<double*>(zn), static_cast<int32_t*>(nullptr));
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(prep_seq_tile_records_kernel, dim3(grid_for(N * int64_t(rec_elems))), dim3(256), 0, c->stream, X, N, L, s, pre, 0, D,
                       rec_elems, static_cast<double*>(xr));
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(tvs_grad_transpose_G_kernel, dim3(unsigned((N + 31) / 32), unsigned(Tpad / 32), unsigned(GL)), dim3(32, 8), 0, c->stream,
                       G, Tn, Tpad, N, GL, static_cast<double*>(gt));
    HIPCHK(c, hipGetLastError());
    for (int64_t r0 = 0; r0 < runs; r0 += chunk_runs) {
        const int64_t nr = (runs - r0 < chunk_runs) ? runs - r0 : chunk_runs;
        const int64_t n0 = r0 * run, n1 = (n0 + nr * run < N) ? n0 + nr * run : N;
        TvsGradTileArgs A;
        memset(&A, 0, sizeof(A));
        A.XR = static_cast<const double*>(xr) + n0 * int64_t(rec_elems);
        A.ZL = static_cast<const double*>(zl); A.ZN = static_cast<const double*>(zn);
        A.Gt = static_cast<const double*>(gt) + n0 * int64_t(GL) * Tpad;
        A.fac = fac ? fac + n0 * int64_t(M + 1) : nullptr;
        A.gfp = gfp ? static_cast<double*>(gfp) : nullptr;
        A.aux = aux ? aux + n0 * int64_t(lt) * Tpad : nullptr;
        A.gzp = static_cast<double*>(gzp) + size_t(r0) * gz_stride;
        A.gxp = static_cast<double*>(gxp);
        A.gbp = gbp ? static_cast<double*>(gbp) + r0 * TB * NR : nullptr;
        A.N = n1 - n0; A.Tn = Tn; A.Tpad = Tpad;
        A.L = L; A.d = d; A.kind = p->base_kernel; A.difference = p->difference; A.M = M;
        A.run = int(run); A.rec_elems = rec_elems; A.order = p->order < M ? p->order : M;
        A.p0 = p->base_params[0]; A.p1 = p->base_params[1];
        A.mat_a1 = p->base_kernel == GPSIG_BASE_MATERN12 ? 0.0 : 1.0; A.mat_a2 = p->base_kernel == GPSIG_BASE_MATERN52 ? 1.0 / 3.0 : 0.0; A.mat_g = pre * mat_c;
        HIPCHK(c, fn(A, dim3(unsigned(TB), unsigned(nr), unsigned(NR)), lds, c->stream));
        const int64_t rows = (n1 - n0) * L;
        hipLaunchKernelGGL(tvs_grad_reduce_gx_kernel, dim3(grid_for(rows * d)), dim3(256), 0, c->stream, static_cast<const double*>(gxp), int(TB) * NR, rows,
                           D, d, 1.0 / pre, gX + n0 * int64_t(L) * d);
        HIPCHK(c, hipGetLastError());
        if (fac) {
            const int64_t nf = (n1 - n0) * (M + 1);
            hipLaunchKernelGGL(tvs_grad_reduce_gf_kernel, dim3(grid_for(nf)), dim3(256), 0, c->stream, static_cast<const double*>(gfp), int(TB), nf,
                               gfac + n0 * int64_t(M + 1));
            HIPCHK(c, hipGetLastError());
        }
    }
    hipLaunchKernelGGL(tvs_grad_reduce_gz_kernel, dim3(grid_for(int64_t(lt) * E * d * Tn)), dim3(256), 0, c->stream, static_cast<const double*>(gzp),
                       int(runs), lt, E, D, Tpad, Tn, d, collapse ? 1 : 0, 1.0 / pre, gZ);
    HIPCHK(c, hipGetLastError());
    if (gbp) {
        hipLaunchKernelGGL(tvs_grad_sum_kernel, dim3(1), dim3(64), 0, c->stream, static_cast<const double*>(gbp), TB * runs * NR, gb);
        HIPCHK(c, hipGetLastError());
    }
    *done = true;
    return GPSIG_OK;
}
}  // namespace gpsig
