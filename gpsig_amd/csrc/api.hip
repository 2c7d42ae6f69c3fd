// api.hip -- implementation of the C ABI declared in include/gpsig_hip.h.
//
// Host-side orchestration only: input preparation, kernel-shape selection, task lists, scratch
// memory, HIP-event timing.  All arithmetic happens in the kernels of seq_gram_kernel.hpp and
// aux_kernels.hpp.  There is deliberately no CPU fallback: without a HIP device every entry point
// fails with GPSIG_ERR_HIP.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gpsig_hip.h"
#include "aux_kernels.hpp"
#include "lowrank_kernels.hpp"
#include "lr_fused_args.hpp"
#include "lr_draw_kernels.hpp"
#include "seq_args.hpp"
#include "seq_configs.hpp"
#include "tvs_tile_kernel.hpp"
#include "seq_pk2_kernel.hpp"
#include "sig_feat_kernel.hpp"

namespace gpsig {
typedef hipError_t (*TvsTileLaunchFn)(TvsTileArgs&, size_t, hipStream_t, int);
TvsTileLaunchFn tvs_tile_lookup(int M, int NW, int D, bool incr, int kind);
TvsTileLaunchFn tvs_tile_lookup_ho(int M, int NW, int D, bool incr);
int tvs_tile_width(int d);
bool seq_pk2_select(int rows, int d, int M, int* G, int* C, int* D);
typedef hipError_t (*SigFeatLaunchFn)(const SigFeatArgs&, unsigned, size_t, hipStream_t);
SigFeatLaunchFn sig_feat_lookup(int d, int M);
hipError_t sig_gram_launch(const SigGramArgs& G, int ntiles, hipStream_t stream, int dma, int* used_dma);
hipError_t sig_reduce_launch(const SigReduceArgs& R, hipStream_t stream);
hipError_t sig_convert_launch(const void* in, void* out, int64_t n, bool widen, hipStream_t stream);
bool solver_dsyevd(void** handle_slot, hipStream_t stream, int n, double* A, double* ev, double* work, int* info, std::string* err);
void solver_release(void* handle);
bool solver_dgemm(void** handle_slot, hipStream_t stream, bool transA, bool transB, int m, int n, int k, double alpha, const double* A, int lda,
                  const double* B, int ldb, double beta, double* C, int ldc, std::string* err);      // lowrank_solver.hip
int tvs_tile_waves(int M, int D, int E, int kind);
typedef hipError_t (*SeqLaunchFn)(const SeqGramArgs&, int, size_t, hipStream_t);
SeqLaunchFn seq_pk2_lookup(int G, int C, int D, int M, int mode, int pack, int waves);
SeqLaunchFn seq_lookup_inc_exact(int, int, int, int, bool);
SeqLaunchFn seq_lookup_inc_ex_g16_d4(int, int, int, int, bool);
SeqLaunchFn seq_lookup_inc_ex_g16_d8(int, int, int, int, bool);
SeqLaunchFn seq_lookup_inc_ex_g16_d16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_inc_ex_g64_d4(int, int, int, int, bool);
SeqLaunchFn seq_lookup_inc_ex_g64_d8(int, int, int, int, bool);
SeqLaunchFn seq_lookup_inc_ex_g64_d16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptd_ex_g16_d4(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptd_ex_g16_d8(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptd_ex_g16_d16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptd_ex_g64_d4(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptd_ex_g64_d8(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptd_ex_g64_d16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_inc_g16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_inc_g64(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptd_exact(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptdrbf_exact(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptdrbf_stash(int, int, int, int);
SeqLaunchFn seq_lookup_ptdmatern_stash(int, int, int, int, int);
SeqLaunchFn seq_lookup_ptdm12_exact(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptdm32_exact(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptdm52_exact(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptdrbf_ex_g16_d4(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptdrbf_ex_g16_d8(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptdrbf_ex_g16_d16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptdrbf_ex_g64_d4(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptdrbf_ex_g64_d8(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptdrbf_ex_g64_d16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptd_g16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptd_spectral_g16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptd_spectral_g64(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptd_g64(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptn_g16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ptn_g64(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ho_inc_d4(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_inc_d8(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_inc_d16(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_ptd_d4(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_ptd_d8(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_ptd_d16(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_f32_inc_d4(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_f32_inc_d8(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_f32_inc_d16(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_f32_ptd_d4(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_f32_ptd_d8(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_f32_ptd_d16(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_ptn_d4(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_ptn_d8(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_ptn_d16(int, int, int, int, int);
typedef hipError_t (*TvsLaunchFn)(const TvsArgs&, hipStream_t);
TvsLaunchFn tvs_lookup(int M, int TT, bool incr, bool f32);
SeqLaunchFn seq_lookup_f32_inc_exact(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_inc_ex_g16_d4(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_inc_ex_g16_d8(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_inc_ex_g16_d16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_inc_ex_g64_d4(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_inc_ex_g64_d8(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_inc_ex_g64_d16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptd_ex_g16_d4(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptd_ex_g16_d8(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptd_ex_g16_d16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptd_ex_g64_d4(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptd_ex_g64_d8(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptd_ex_g64_d16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_inc_g16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_inc_g64(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptd_exact(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptdrbf_exact(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptdrbf_ex_g16_d4(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptdrbf_ex_g16_d8(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptdrbf_ex_g16_d16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptdrbf_ex_g64_d4(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptdrbf_ex_g64_d8(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptdrbf_ex_g64_d16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptd_g16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptd_g64(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptn_g16(int, int, int, int, bool);
SeqLaunchFn seq_lookup_f32_ptn_g64(int, int, int, int, bool);
SeqLaunchFn seq_lookup_ho_f32_ptn_d4(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_f32_ptn_d8(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_f32_ptn_d16(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_inc_d32(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_ptd_d32(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_ptn_d32(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_f32_inc_d32(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_f32_ptd_d32(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_f32_ptn_d32(int, int, int, int, int);
SeqLaunchFn seq_lookup_ho_ptdrbf_exact(int G, int C, int D, int M, int order);
SeqLaunchFn seq_lookup_ho_ptdrbf_exact_o4(int G, int C, int D, int M, int order);
SeqLaunchFn seq_lookup_ho_ptdm12_exact(int kind, int G, int C, int D, int M, int order);     // the Matern families' exact higher-order instances
SeqLaunchFn seq_lookup_ho_ptdm32_exact(int kind, int G, int C, int D, int M, int order);
SeqLaunchFn seq_lookup_ho_ptdm52_exact(int kind, int G, int C, int D, int M, int order);
typedef hipError_t (*TvsLaneTLaunchFn)(const TvsLaneTArgs&, hipStream_t);
bool tvs_lanet_plan(int M, int d, bool incr, TvsLaneTLaunchFn* fns, int* ngroups);
// wide_api.hip: state spaces beyond the exact-shape kernels' columns (kernel arguments by dgemm, fused map / difference / recursion kernels)
bool wide_tvs_available(const gpsig_ctx* c, const gpsig_params* p, int d, int64_t Tn, int64_t N, int L);
int wide_tvs_forward(gpsig_ctx* c, const gpsig_params* p, const ScaleParams& sz, int d, const double* Z, const double* Xs, int64_t Tn, int64_t N, int L,
                     int increments, const double* fx, const double* w, int sum_levels, double* out, double* aux);
bool wide_tens_available(const gpsig_ctx* c, const gpsig_params* p, int64_t Tn);
int wide_tens_forward(gpsig_ctx* c, const gpsig_params* p, const ScaleParams& sz, int d, const double* Z, int64_t Tn, int increments, const double* w,
                      int sum_levels, double* out);
bool wide_lat_available(const gpsig_ctx* c, const gpsig_params* p, int L1, int L2);
int wide_lat_forward(gpsig_ctx* c, const gpsig_params* p, int d, const double* Xs, const double* Ys, int64_t N1, int64_t N2, int L1, int L2, bool diag,
                     double* out);
}  // namespace gpsig

using namespace gpsig;

namespace {

#define X_CFG(G_, C_, D_, MM_, EX_) {G_, C_, D_, MM_, EX_},
const SeqConfig SEQ_TABLE[] = {GPSIG_SEQ_CONFIGS_ALL(X_CFG)};
const SeqConfig SEQ_TABLE_GENERIC[] = {GPSIG_SEQ_CONFIGS_GENERIC(X_CFG)};
const SeqConfig SEQ_TABLE_SPECTRAL[] = {GPSIG_SEQ_CONFIGS_SPECTRAL(X_CFG)};
#undef X_CFG
#define X_HO(G_, C_, D_, MM_, OM_) {G_, C_, D_, MM_, OM_},
const SeqHOConfig SEQ_HO_TABLE[] = {GPSIG_SEQ_HO_ALL(X_HO)};
#undef X_HO
constexpr int N_SEQ_HO_TABLE = int(sizeof(SEQ_HO_TABLE) / sizeof(SEQ_HO_TABLE[0]));
#define X_CFG2(G_, C_, D_, MM_, EX_) {G_, C_, D_, MM_, EX_},
const SeqConfig SEQ_TABLE_F32[] = {GPSIG_SEQ_CONFIGS_F32_ALL(X_CFG2)};
#undef X_CFG2
constexpr int N_SEQ_TABLE_F32 = int(sizeof(SEQ_TABLE_F32) / sizeof(SEQ_TABLE_F32[0]));
constexpr int N_SEQ_TABLE = int(sizeof(SEQ_TABLE) / sizeof(SEQ_TABLE[0]));
constexpr int N_SEQ_TABLE_GENERIC = int(sizeof(SEQ_TABLE_GENERIC) / sizeof(SEQ_TABLE_GENERIC[0]));
constexpr int N_SEQ_TABLE_SPECTRAL = int(sizeof(SEQ_TABLE_SPECTRAL) / sizeof(SEQ_TABLE_SPECTRAL[0]));

// float64, differences, the RBF kernel at compile time.  These instances use the table-driven exp on prescaled records
// (SEQ_FAST_RBF in seq_gram_kernel.hpp): whoever launches one prepares the records with the prescale and the norm column.
SeqLaunchFn seq_launcher_rbf(const SeqConfig& c) {
    SeqLaunchFn f = nullptr;
    if ((f = seq_lookup_ptdrbf_exact(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
    if ((f = seq_lookup_ptdrbf_ex_g16_d4(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
    if ((f = seq_lookup_ptdrbf_ex_g16_d8(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
    if ((f = seq_lookup_ptdrbf_ex_g16_d16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
    if ((f = seq_lookup_ptdrbf_ex_g64_d4(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
    if ((f = seq_lookup_ptdrbf_ex_g64_d8(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
    return seq_lookup_ptdrbf_ex_g64_d16(c.G, c.C, c.D, c.MMAX, c.exact);
}

SeqLaunchFn seq_launcher(int mode, const SeqConfig& c, bool f32, int kind) {
    SeqLaunchFn f = nullptr;
    if (f32) {
        if (mode == MODE_INC) {
            if ((f = seq_lookup_f32_inc_exact(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (c.exact && (f = seq_lookup_f32_inc_ex_g16_d4(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (c.exact && (f = seq_lookup_f32_inc_ex_g16_d8(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (c.exact && (f = seq_lookup_f32_inc_ex_g16_d16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (c.exact && (f = seq_lookup_f32_inc_ex_g64_d4(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (c.exact && (f = seq_lookup_f32_inc_ex_g64_d8(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (c.exact && (f = seq_lookup_f32_inc_ex_g64_d16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if ((f = seq_lookup_f32_inc_g16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            return seq_lookup_f32_inc_g64(c.G, c.C, c.D, c.MMAX, c.exact);
        }
        if (mode == MODE_PT_DIFF) {
            if (kind == BASE_RBF && c.exact && (f = seq_lookup_f32_ptdrbf_exact(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (kind == BASE_RBF && c.exact && (f = seq_lookup_f32_ptdrbf_ex_g16_d4(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (kind == BASE_RBF && c.exact && (f = seq_lookup_f32_ptdrbf_ex_g16_d8(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (kind == BASE_RBF && c.exact && (f = seq_lookup_f32_ptdrbf_ex_g16_d16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (kind == BASE_RBF && c.exact && (f = seq_lookup_f32_ptdrbf_ex_g64_d4(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (kind == BASE_RBF && c.exact && (f = seq_lookup_f32_ptdrbf_ex_g64_d8(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (kind == BASE_RBF && c.exact && (f = seq_lookup_f32_ptdrbf_ex_g64_d16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if ((f = seq_lookup_f32_ptd_exact(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (c.exact && (f = seq_lookup_f32_ptd_ex_g16_d4(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (c.exact && (f = seq_lookup_f32_ptd_ex_g16_d8(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (c.exact && (f = seq_lookup_f32_ptd_ex_g16_d16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (c.exact && (f = seq_lookup_f32_ptd_ex_g64_d4(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (c.exact && (f = seq_lookup_f32_ptd_ex_g64_d8(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if (c.exact && (f = seq_lookup_f32_ptd_ex_g64_d16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            if ((f = seq_lookup_f32_ptd_g16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
            return seq_lookup_f32_ptd_g64(c.G, c.C, c.D, c.MMAX, c.exact);
        }
        if (c.exact) return nullptr;
        if ((f = seq_lookup_f32_ptn_g16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        return seq_lookup_f32_ptn_g64(c.G, c.C, c.D, c.MMAX, c.exact);
    }
    if (mode == MODE_INC) {
        if ((f = seq_lookup_inc_exact(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if (c.exact && (f = seq_lookup_inc_ex_g16_d4(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if (c.exact && (f = seq_lookup_inc_ex_g16_d8(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if (c.exact && (f = seq_lookup_inc_ex_g16_d16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if (c.exact && (f = seq_lookup_inc_ex_g64_d4(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if (c.exact && (f = seq_lookup_inc_ex_g64_d8(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if (c.exact && (f = seq_lookup_inc_ex_g64_d16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if ((f = seq_lookup_inc_g16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        return seq_lookup_inc_g64(c.G, c.C, c.D, c.MMAX, c.exact);
    }
    if (mode == MODE_PT_DIFF) {
        if ((f = seq_lookup_ptd_exact(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if (c.exact && (f = seq_lookup_ptd_ex_g16_d4(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if (c.exact && (f = seq_lookup_ptd_ex_g16_d8(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if (c.exact && (f = seq_lookup_ptd_ex_g16_d16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if (c.exact && (f = seq_lookup_ptd_ex_g64_d4(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if (c.exact && (f = seq_lookup_ptd_ex_g64_d8(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if (c.exact && (f = seq_lookup_ptd_ex_g64_d16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        if ((f = seq_lookup_ptd_g16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
        return seq_lookup_ptd_g64(c.G, c.C, c.D, c.MMAX, c.exact);
    }
    if ((f = seq_lookup_ptn_g16(c.G, c.C, c.D, c.MMAX, c.exact))) return f;
    return seq_lookup_ptn_g64(c.G, c.C, c.D, c.MMAX, c.exact);
}

SeqLaunchFn seq_launcher_ho(int mode, const SeqHOConfig& c, bool f32) {
#define HO_PICK(V)                                                                \
    do {                                                                          \
        if (c.D == 4) return seq_lookup_ho_##V##_d4(c.G, c.C, c.D, c.MMAX, c.OMAX);   \
        if (c.D == 8) return seq_lookup_ho_##V##_d8(c.G, c.C, c.D, c.MMAX, c.OMAX);   \
        if (c.D == 16) return seq_lookup_ho_##V##_d16(c.G, c.C, c.D, c.MMAX, c.OMAX); \
        return seq_lookup_ho_##V##_d32(c.G, c.C, c.D, c.MMAX, c.OMAX);            \
    } while (0)
    if (f32) {
        if (mode == MODE_INC) HO_PICK(f32_inc);
        if (mode == MODE_PT_DIFF) HO_PICK(f32_ptd);
        HO_PICK(f32_ptn);
    }
    if (mode == MODE_INC) HO_PICK(inc);
    if (mode == MODE_PT_DIFF) HO_PICK(ptd);
    HO_PICK(ptn);
#undef HO_PICK
}

}  // namespace

#include <new>

#include "ctx.hpp"

namespace {

int check_params(gpsig_ctx* c, const gpsig_params* p) {
    if (!c) return GPSIG_ERR_INVALID;
    if (!p) return fail(c, GPSIG_ERR_INVALID, "params is NULL");
    if (p->dtype != GPSIG_F64 && p->dtype != GPSIG_F32) return fail(c, GPSIG_ERR_INVALID, "unknown dtype %d", p->dtype);
    if (p->num_levels < 1) return fail(c, GPSIG_ERR_INVALID, "num_levels must be >= 1");
    if (p->num_features < 1 || p->num_features > MAX_FEATURES_WIDE)
        return fail(c, GPSIG_ERR_UNSUPPORTED, "num_features=%d outside [1, %d]", p->num_features, MAX_FEATURES_WIDE);
    if (p->num_lags < 0 || p->num_lags > MAX_LAGS) return fail(c, GPSIG_ERR_UNSUPPORTED, "num_lags=%d outside [0, %d]", p->num_lags, MAX_LAGS);
    if (p->base_kernel < GPSIG_BASE_LINEAR || p->base_kernel > GPSIG_BASE_SPECTRAL) return fail(c, GPSIG_ERR_INVALID, "unknown base kernel %d", p->base_kernel);
    if (p->base_kernel == GPSIG_BASE_SPECTRAL) {
        const int Q = int(p->base_params[0]), fam = int(p->base_params[1]);
        if (Q < 1 || Q > 64 || fam < 0 || fam > 2) return fail(c, GPSIG_ERR_INVALID, "spectral kernel: bad number of components / family");
        if (p->num_features > SPECTRAL_STRIDE) return fail(c, GPSIG_ERR_UNSUPPORTED, "the spectral base kernel is built for at most %d features", int(SPECTRAL_STRIDE));
        if (!p->base_table || p->base_table_len != int64_t(Q) * (1 + 2 * p->num_features))
            return fail(c, GPSIG_ERR_INVALID, "spectral kernel: base_table must hold alpha[Q], omega[Q][d], gamma[Q][d]");
        if (p->lengthscales || p->num_lags != 0)
            return fail(c, GPSIG_ERR_INVALID, "spectral kernel: lengthscales must be NULL and num_lags 0 (kernels.py:907, :82 vs :913)");
        if (p->dtype != GPSIG_F64) return fail(c, GPSIG_ERR_UNSUPPORTED, "the spectral base kernel is built for float64 only");
    }
    if (p->order < 1 || p->order > p->num_levels) return fail(c, GPSIG_ERR_INVALID, "order=%d outside [1, num_levels]", p->order);
    if (!p->variances) return fail(c, GPSIG_ERR_INVALID, "variances is NULL");
    if (p->num_lags > 0 && (!p->lags || !p->gamma)) return fail(c, GPSIG_ERR_INVALID, "num_lags > 0 needs lags and gamma");
    return GPSIG_OK;
}

ScaleParams scale_of(const gpsig_params* p, bool apply_scaling) {
    ScaleParams s;
    memset(&s, 0, sizeof(s));
    s.d_in = p->num_features;
    s.num_lags = apply_scaling ? p->num_lags : 0;
    s.has_ls = apply_scaling && p->lengthscales != nullptr;
    for (int f = 0; f < p->num_features && f < MAX_FEATURES; ++f) s.ls[f] = s.has_ls ? p->lengthscales[f] : 1.0;
    for (int l = 0; l < s.num_lags; ++l) s.lags[l] = p->lags[l];
    for (int l = 0; l <= s.num_lags; ++l) s.gamma[l] = s.num_lags > 0 ? p->gamma[l] : 1.0;
    s.jitter = 1e-6;   // settings.jitter inside lin_interp (gpsig/lags.py:22)
    return s;
}

// The same for a launch: state spaces wider than MAX_FEATURES read their lengthscales from a device copy (cached by content)
int scale_params(gpsig_ctx* c, const gpsig_params* p, bool apply_scaling, ScaleParams* out) {
    *out = scale_of(p, apply_scaling);
    if (!out->has_ls || p->num_features <= MAX_FEATURES) return GPSIG_OK;
    const size_t n = size_t(p->num_features);
    void* d;
    CHK(ensure(c, B_LS, sizeof(double) * n, &d));
    if (!(c->ls_base == d && c->last_ls.size() == n && memcmp(c->last_ls.data(), p->lengthscales, sizeof(double) * n) == 0)) {
        CHK(no_capture(c, "the lengthscales of a wide state space changed and have to be uploaded"));
        ++c->alloc_gen;                  // a recorded graph read the old contents of this buffer: its replays are refused from here on
        c->ls_base = nullptr;
        c->last_ls.assign(p->lengthscales, p->lengthscales + n);
        HIPCHK(c, hipMemcpyAsync(d, c->last_ls.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        CHK(host_sync(c));
        c->ls_base = d;
    }
    out->ls_dev = static_cast<const double*>(d);
    return GPSIG_OK;
}

// BASE_SPECTRAL: alpha[Q], omega[Q][SPECTRAL_STRIDE], gamma[Q][SPECTRAL_STRIDE] on the device (zero padded); NULL otherwise
int spectral_table(gpsig_ctx* c, const gpsig_params* p, const double** dev) {
    *dev = nullptr;
    if (p->base_kernel != GPSIG_BASE_SPECTRAL) return GPSIG_OK;
    const int Q = int(p->base_params[0]), d = p->num_features;
    std::vector<double> h(size_t(Q) * (1 + 2 * SPECTRAL_STRIDE), 0.0);
    for (int q = 0; q < Q; ++q) {
        h[q] = p->base_table[q];
        for (int f = 0; f < d; ++f) {
            h[Q + size_t(q) * SPECTRAL_STRIDE + f] = p->base_table[Q + size_t(q) * d + f];
            h[Q + size_t(Q) * SPECTRAL_STRIDE + size_t(q) * SPECTRAL_STRIDE + f] = p->base_table[Q + size_t(Q) * d + size_t(q) * d + f];
        }
    }
    void* dp;
    CHK(ensure(c, B_SPEC, sizeof(double) * h.size(), &dp));
    if (!(c->spec_base == dp && c->last_spec == h)) {       // unchanged parameters: the device copy is still right (as the level weights)
        CHK(no_capture(c, "the spectral kernel's table changed"));
        ++c->alloc_gen;                  // a recorded graph read the old contents of this buffer: its replays are refused from here on
        c->spec_base = nullptr;
        HIPCHK(c, hipMemcpyAsync(dp, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, c->stream));
        CHK(host_sync(c));
        c->spec_base = dp;
        c->last_spec = h;
    }
    *dev = static_cast<const double*>(dp);
    return GPSIG_OK;
}

void base_p(const gpsig_params* p, double* p0, double* p1) {
    *p0 = p->base_params[0];
    *p1 = p->base_params[1];
}

// weights sigma * variances[m] on the device
int upload_weights(gpsig_ctx* c, const gpsig_params* p, const double** w) {
    const int M1 = p->num_levels + 1;
    std::vector<double> h(M1);
    for (int m = 0; m < M1; ++m) h[m] = p->sigma * p->variances[m];
    void* d;
    CHK(ensure(c, B_W, sizeof(double) * M1, &d));
    if (c->last_weights != h) {          // unchanged hyper-parameters: the device copy is still right
        CHK(no_capture(c, "the level weights changed"));
        ++c->alloc_gen;                  // a recorded graph was made for the old weights
        c->last_weights.clear();
        HIPCHK(c, hipMemcpyAsync(d, h.data(), sizeof(double) * M1, hipMemcpyHostToDevice, c->stream));
        CHK(host_sync(c));   // h goes out of scope
        c->last_weights = h;
    }
    *w = static_cast<const double*>(d);
    return GPSIG_OK;
}

#define ENTER(c, p)                         \
    CHK(check_params((c), (p)));            \
    HIPCHK((c), hipSetDevice((c)->device));


// ---- low-rank mode (float64) ---------------------------------------------------------------------------
struct LrDev {                 // device copies of a gpsig_lowrank
    const double* S; const double* Wh;
    int c, r, nsk;
    std::vector<const int32_t*> colptr, i1, i2;
    std::vector<const double*> val;
    std::vector<const LrEntry*> ent;        // the same entries packed (value, i1, i2) for the fused feature kernel
    std::vector<int> k1, k2;
};

// entries a projection is given room for: the expected number plus twelve standard deviations (never more than all of them)
int32_t lr_sketch_capacity(int64_t D, int r, int sparsity) {
    if (sparsity == 2) return int32_t(r);
    const double s = sparsity == 0 ? sqrt(double(D)) : double(D) / log(double(D));
    const double mean = double(D) * r / (s < 1.0 ? 1.0 : s);
    double cap = mean + 12.0 * sqrt(mean) + 64.0;
    if (cap > double(D) * r) cap = double(D) * r;
    return int32_t(cap);
}

int lr_state_layout(gpsig_lr_state* st, int M) {
    using gpsig::LrEntry;
    size_t o = 0;
    auto take = [&](size_t n, size_t align = 16) { o = (o + align - 1) / align * align; const size_t at = o; o += n; return at; };
    const int c = st->c;
    const size_t o_idx = take(sizeof(int64_t) * (size_t(c) + size_t(st->r) + 8));
    const size_t o_S = take(sizeof(double) * size_t(c) * st->d_eff);
    const size_t o_jd = take(sizeof(double) * c), o_W = take(sizeof(double) * size_t(c) * c), o_Wh = take(sizeof(double) * size_t(c) * c);
    const size_t o_WhT = take(sizeof(double) * size_t(c) * c), o_ev = take(sizeof(double) * c), o_work = take(sizeof(double) * c);
    const size_t o_info = take(sizeof(int) * 4);
    size_t o_sk[gpsig::LR_FUSED_MAX_SKETCHES][6];
    int k2 = c;
    for (int i = 0; i < st->nsk; ++i) {
        gpsig_lr_state::Sk& s = st->sk[i];
        s.k1 = c; s.k2 = k2;
        s.cap = lr_sketch_capacity(int64_t(c) * k2, st->r, st->sparsity);
        o_sk[i][0] = take(sizeof(int32_t) * (size_t(st->r) + 1));
        o_sk[i][1] = take(sizeof(int32_t) * (size_t(st->r) + 1));
        o_sk[i][2] = take(sizeof(int32_t) * size_t(s.cap));
        o_sk[i][3] = take(sizeof(int32_t) * size_t(s.cap));
        o_sk[i][4] = take(sizeof(double) * size_t(s.cap));
        o_sk[i][5] = take(sizeof(LrEntry) * (size_t(s.cap) + 1));
        k2 = st->r;
    }
    (void)M;
    if (o > st->bytes) {
        if (st->block) { (void)hipStreamSynchronize(st->ctx->stream); (void)hipFree(st->block); st->block = nullptr; st->bytes = 0; }
        if (hipMalloc(&st->block, o + 256) != hipSuccess) return GPSIG_ERR_NOMEM;
        st->bytes = o + 256;
    }
    char* b = static_cast<char*>(st->block);
    st->idx = reinterpret_cast<int64_t*>(b + o_idx);
    st->S = reinterpret_cast<double*>(b + o_S); st->jd = reinterpret_cast<double*>(b + o_jd); st->W = reinterpret_cast<double*>(b + o_W);
    st->Wh = reinterpret_cast<double*>(b + o_Wh); st->WhT = reinterpret_cast<double*>(b + o_WhT); st->ev = reinterpret_cast<double*>(b + o_ev);
    st->work = reinterpret_cast<double*>(b + o_work); st->info = reinterpret_cast<int*>(b + o_info);
    for (int i = 0; i < st->nsk; ++i) {
        gpsig_lr_state::Sk& s = st->sk[i];
        s.counts = reinterpret_cast<int32_t*>(b + o_sk[i][0]); s.colptr = reinterpret_cast<int32_t*>(b + o_sk[i][1]);
        s.i1 = reinterpret_cast<int32_t*>(b + o_sk[i][2]); s.i2 = reinterpret_cast<int32_t*>(b + o_sk[i][3]);
        s.val = reinterpret_cast<double*>(b + o_sk[i][4]); s.ent = reinterpret_cast<LrEntry*>(b + o_sk[i][5]);
    }
    return GPSIG_OK;
}

// the LrDev view of a device-resident state (what lr_upload builds from host arrays)
void lr_state_dev(const gpsig_lr_state* st, LrDev* D) {
    D->c = st->c; D->r = st->r; D->nsk = st->nsk;
    D->S = st->S; D->Wh = st->Wh;
    for (int i = 0; i < st->nsk; ++i) {
        D->colptr.push_back(st->sk[i].colptr); D->i1.push_back(st->sk[i].i1); D->i2.push_back(st->sk[i].i2);
        D->val.push_back(st->sk[i].val); D->ent.push_back(st->sk[i].ent);
        D->k1.push_back(st->sk[i].k1); D->k2.push_back(st->sk[i].k2);
    }
}


int lr_check(gpsig_ctx* c, const gpsig_params* p, const gpsig_lowrank* lr) {
    if (!lr) return fail(c, GPSIG_ERR_INVALID, "lowrank descriptor is NULL");
    if (p->dtype != GPSIG_F64) return fail(c, GPSIG_ERR_UNSUPPORTED, "low-rank mode is built for float64 only");
    if (p->base_kernel == GPSIG_BASE_SPECTRAL) return fail(c, GPSIG_ERR_UNSUPPORTED, "low-rank mode is not built for the spectral base kernel");
    if (p->order != 1 && p->num_levels > 1) return fail(c, GPSIG_ERR_UNSUPPORTED, "Low-rank mode not implemented for order higher than 1.");
    if (lr->num_components < 1 || lr->rank_bound < 1) return fail(c, GPSIG_ERR_INVALID, "num_components and rank_bound must be positive");
    if (lr->num_sketches != p->num_levels - 1) return fail(c, GPSIG_ERR_INVALID, "need one sketch per level 2..num_levels");
    if (lr->device_state) {                  // drawn on the device (gpsig_lr_draw): nothing on the host to check
        const gpsig_lr_state* st = lr->device_state;
        if (st->ctx != c) return fail(c, GPSIG_ERR_INVALID, "the low-rank state belongs to another context");
        if (st->c != lr->num_components || st->r != lr->rank_bound || st->nsk != lr->num_sketches)
            return fail(c, GPSIG_ERR_INVALID, "the low-rank state was drawn for other sizes");
        return GPSIG_OK;
    }
    if (!lr->landmarks || !lr->whitening || (lr->num_sketches > 0 && !lr->sketches)) return fail(c, GPSIG_ERR_INVALID, "NULL low-rank array");
    int k2 = lr->num_components;
    for (int i = 0; i < lr->num_sketches; ++i) {
        const gpsig_sketch& sk = lr->sketches[i];
        if (sk.k1 != lr->num_components || sk.k2 != k2 || sk.r != lr->rank_bound)
            return fail(c, GPSIG_ERR_INVALID, "sketch %d has shape (%d, %d) -> %d, expected (%d, %d) -> %d", i, sk.k1, sk.k2, sk.r,
                        lr->num_components, k2, lr->rank_bound);
        k2 = lr->rank_bound;
    }
    return GPSIG_OK;
}

// FNV-1a over 64-bit words (a tail shorter than a word is folded in bytewise)
static uint64_t lr_fnv(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, b + i, 8);
        h = (h ^ w) * 0x100000001b3ull;
    }
    for (; i < n; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
    return h;
}

// upload landmarks, whitening and sketches into one scratch block -- unless the block already holds exactly these (content hash):
// the random objects of one evaluation go through several calls, and every upload is a host synchronisation
int lr_upload(gpsig_ctx* c, const gpsig_params* p, const gpsig_lowrank* lr, int d_eff, LrDev* D) {
    if (lr->device_state) {                  // already where the kernels read it
        if (lr->device_state->d_eff != d_eff) return fail(c, GPSIG_ERR_INVALID, "the low-rank state was drawn for %d columns, the call has %d", lr->device_state->d_eff, d_eff);
        lr_state_dev(lr->device_state, D);
        return GPSIG_OK;
    }
    const int cc = lr->num_components;
    uint64_t hsh = 0xcbf29ce484222325ull;
    const int64_t dims[4] = {cc, d_eff, lr->rank_bound, lr->num_sketches};
    hsh = lr_fnv(hsh, dims, sizeof(dims));
    hsh = lr_fnv(hsh, lr->landmarks, sizeof(double) * size_t(cc) * d_eff);
    hsh = lr_fnv(hsh, lr->whitening, sizeof(double) * size_t(cc) * cc);
    for (int i = 0; i < lr->num_sketches; ++i) {
        const gpsig_sketch& sk = lr->sketches[i];
        const int64_t sd[4] = {sk.k1, sk.k2, sk.r, sk.nnz};
        hsh = lr_fnv(hsh, sd, sizeof(sd));
        hsh = lr_fnv(hsh, sk.colptr, sizeof(int32_t) * (size_t(sk.r) + 1));
        hsh = lr_fnv(hsh, sk.i1, sizeof(int32_t) * size_t(sk.nnz));
        hsh = lr_fnv(hsh, sk.i2, sizeof(int32_t) * size_t(sk.nnz));
        hsh = lr_fnv(hsh, sk.val, sizeof(double) * size_t(sk.nnz));
    }
    if (hsh == 0) hsh = 1;
    size_t bytes = sizeof(double) * (size_t(cc) * d_eff + size_t(cc) * cc);
    for (int i = 0; i < lr->num_sketches; ++i)
        bytes += sizeof(int32_t) * (size_t(lr->sketches[i].r) + 1 + 2 * size_t(lr->sketches[i].nnz)) + sizeof(double) * size_t(lr->sketches[i].nnz) +
                 sizeof(LrEntry) * size_t(lr->sketches[i].nnz) + 96;
    void* dbase;
    CHK(ensure(c, B_LR0, bytes + 64, &dbase));
    const bool cached = c->lr_hash == hsh && c->lr_base == dbase && c->lr_offsets.size() == size_t(2 + 5 * lr->num_sketches);
    std::vector<unsigned char> h(cached ? 0 : bytes + 64);
    std::vector<size_t> offs;
    size_t o = 0;
    auto place = [&](const void* src, size_t n, size_t align = 8) -> unsigned char* {
        size_t at;
        if (cached) {
            at = c->lr_offsets[offs.size()];
        } else {
            o = (o + align - 1) / align * align;
            if (n) memcpy(h.data() + o, src, n);
            at = o;
            o += n;
        }
        offs.push_back(at);
        return static_cast<unsigned char*>(dbase) + at;
    };
    D->c = cc; D->r = lr->rank_bound; D->nsk = lr->num_sketches;
    D->S = reinterpret_cast<const double*>(place(lr->landmarks, sizeof(double) * size_t(cc) * d_eff));
    D->Wh = reinterpret_cast<const double*>(place(lr->whitening, sizeof(double) * size_t(cc) * cc));
    for (int i = 0; i < lr->num_sketches; ++i) {
        const gpsig_sketch& sk = lr->sketches[i];
        D->colptr.push_back(reinterpret_cast<const int32_t*>(place(sk.colptr, sizeof(int32_t) * (size_t(sk.r) + 1))));
        D->i1.push_back(reinterpret_cast<const int32_t*>(place(sk.i1, sizeof(int32_t) * size_t(sk.nnz))));
        D->i2.push_back(reinterpret_cast<const int32_t*>(place(sk.i2, sizeof(int32_t) * size_t(sk.nnz))));
        D->val.push_back(reinterpret_cast<const double*>(place(sk.val, sizeof(double) * size_t(sk.nnz))));
        std::vector<LrEntry> packed;
        if (!cached) {
            packed.resize(size_t(sk.nnz) + 1);
            for (int64_t e = 0; e < sk.nnz; ++e) packed[size_t(e)] = LrEntry{sk.val[e], sk.i1[e], sk.i2[e]};
        }
        D->ent.push_back(reinterpret_cast<const LrEntry*>(place(packed.data(), sizeof(LrEntry) * size_t(sk.nnz), 16)));
        D->k1.push_back(sk.k1); D->k2.push_back(sk.k2);
    }
    if (!cached) {
        CHK(no_capture(c, "the low-rank objects changed and have to be uploaded"));
        ++c->alloc_gen;                  // a recorded graph read the old contents of this buffer: its replays are refused from here on
        c->lr_hash = 0;
        HIPCHK(c, hipMemcpyAsync(dbase, h.data(), o, hipMemcpyHostToDevice, c->stream));
        CHK(host_sync(c));                       // h goes out of scope
        c->lr_hash = hsh; c->lr_base = dbase; c->lr_offsets = offs;
    }
    (void)p;
    return GPSIG_OK;
}

int lr_gemm(gpsig_ctx* c, const double* A, const double* B, int64_t N1, int64_t N2, int K, int64_t lda, int64_t ldb, double* C_,
            int64_t ldc) {
    if (N1 <= 0 || N2 <= 0) return GPSIG_OK;
    if (c->lr_gemm != 0 && N1 * N2 >= int64_t(GEMM_BM) * GEMM_BN) {       // 128 x 128 tiles through LDS
        dim3 grid((unsigned)((N2 + GEMM_BN - 1) / GEMM_BN), (unsigned)((N1 + GEMM_BM - 1) / GEMM_BM));
        hipLaunchKernelGGL(gemm_abt_f64_tiled_kernel, grid, dim3(256), 0, c->stream, A, B, N1, N2, K, lda, ldb, C_, ldc);
        HIPCHK(c, hipGetLastError());
        return GPSIG_OK;
    }
    dim3 grid((unsigned)((N2 + 63) / 64), (unsigned)((N1 + 63) / 64));
    hipLaunchKernelGGL(gemm_abt_f64_mfma_kernel, grid, dim3(256), 0, c->stream, A, B, N1, N2, K, lda, ldb, C_, ldc);
    HIPCHK(c, hipGetLastError());
    return GPSIG_OK;
}

int lr_level_offsets(gpsig_ctx* c, int M, int cc, int r, const int32_t** dev_off, int* F) {
    std::vector<int32_t> off(M + 2);
    off[0] = 0; off[1] = 1;
    if (M >= 1) off[2] = 1 + cc;
    for (int m = 2; m <= M; ++m) off[m + 1] = off[m] + r;
    *F = off[M + 1];
    void* d;
    CHK(ensure(c, B_LR1, sizeof(int32_t) * off.size(), &d));
    if (!(c->lr_off_base == d && c->lr_off_key[0] == M && c->lr_off_key[1] == cc && c->lr_off_key[2] == r)) {     // (M, c, r) define them
        CHK(no_capture(c, "the low-rank level offsets have to be uploaded"));
        ++c->alloc_gen;                  // a recorded graph read the old contents of this buffer: its replays are refused from here on
        c->lr_off_base = nullptr;
        HIPCHK(c, hipMemcpyAsync(d, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice, c->stream));
        CHK(host_sync(c));
        c->lr_off_base = d; c->lr_off_key[0] = M; c->lr_off_key[1] = cc; c->lr_off_key[2] = r;
    }
    *dev_off = static_cast<const int32_t*>(d);
    return GPSIG_OK;
}

constexpr size_t TIMING_MAX_EVENTS = 8192;
int timing_begin_any(gpsig_ctx* c, hipEvent_t* e0, hipEvent_t* e1, bool* on) {
    *on = false;
    if (c->capturing || c->ev_used + 2 > TIMING_MAX_EVENTS) return GPSIG_OK;
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


// ---- the linear / cosine kernel's Kzx as a product of level features (round 4) ------------------------------------------------------
// An inducing tensor's level m is the rank-one tensor z_1 (x) .. (x) z_m (first factor <-> earliest time, signature_algs.py:118-125), and
// K_m(z, x) = <z_1 (x) .. (x) z_m, Phi_m(x)> with the level features of sig_feat_kernel.hpp (every order: :129-160 too).
// Zf[t][k]: the tensors' features in the layout of the sequences' (natural order, level-0 column = 1, zero padding).  ZT / ZS as
// prep_tensors_kernel leaves them: ZT[((t * d + f) * lt + c) * E + e], ZS[(t * lt + c) * E + e] = |z|^2.
static __global__ void tens_level_features_kernel(const double* __restrict__ ZT, const double* __restrict__ ZS, int lt, int E, int d, int64_t Tn,
                                                  int M, int unit, int64_t ld, double* __restrict__ Zf) {
    const int64_t total = Tn * ld;
    for (int64_t e = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t t = e / ld;
        int k = int(e - t * ld), m = 1, w = d, off = 0;
        while (m <= M && k >= off + w) { off += w; ++m; w *= d; }
        double v;
        if (m > M) {
            v = k == off ? 1.0 : 0.0;                                 // level 0, then the padding
        } else {
            int idx = k - off;
            const int c0 = m * (m - 1) / 2;
            v = 1.0;
            for (int j = m - 1; j >= 0; --j) {                        // last index fastest = the component paired with the latest time
                const int f = idx % d;
                idx /= d;
                const int c = c0 + j;
                double z = ZT[((t * d + f) * lt + c) * E + (E - 1)];
                if (unit) z *= rsqrt(ZS[(t * lt + c) * E + (E - 1)]);  // SignatureCosine: unit vectors (kernels.py:820-828)
                if (E == 2) {                                         // increments: k(z1, x) - k(z0, x) is linear in z (kernels.py:329-330)
                    double z0 = ZT[((t * d + f) * lt + c) * E];
                    if (unit) z0 *= rsqrt(ZS[(t * lt + c) * E]);
                    z -= z0;
                }
                v *= z;
            }
        }
        Zf[e] = v;
    }
}

static __global__ void square_small_kernel(const double* __restrict__ a, int n, double* __restrict__ b) {
    if (int(threadIdx.x) < n) b[threadIdx.x] = a[threadIdx.x] * a[threadIdx.x];
}

// ---- SignatureLinear, first order: the Gram as a contraction of explicit level features (sig_feat_kernel.hpp) --------------------
// Taken where it is the cheaper evaluation -- 2 sum_m d^m flops per entry on the matrix cores against the lattice sweep's
// L1 L2 (2d + 3M - 1) on the vector unit at about 0.6 of the GEMM's efficiency -- and the feature matrices fit.  *done = false
// leaves the call to the lattice kernels.  row_end > 0: the owned entries of rows [row_begin, row_end) (multi-GPU row blocks).
int sig_features_K(gpsig_ctx* c, const gpsig_params* p, bool raw, const void* X, const void* X2, int64_t N1, int64_t N2, int L1, int L2,
                   int return_levels, void* out, bool timed, int x_squared, int64_t row_begin, int64_t row_end, int compact, bool* done,
                   bool row_block_call = false) {
    *done = false;
    // the linear kernel, and the cosine kernel as the linear kernel of the unit vectors x / |x| (kernels.py:820-828)
    const bool cosine = p->base_kernel == GPSIG_BASE_COSINE;
    // float32 calls too: computed in float64 (the matrix cores' own precision), inputs widened and the result rounded -- 29 -> 1.4 ms at
    // d = 16, num_levels = 3, and closer to the reference than a float32 recursion; not for row blocks (float64 only)
    const bool f32 = p->dtype == GPSIG_F32;
    if (c->sig_features == 0 || !(p->dtype == GPSIG_F64 || f32) || !(p->base_kernel == GPSIG_BASE_LINEAR || cosine)) return GPSIG_OK;
    if (x_squared && (X2 == nullptr || raw || !p->normalization)) return GPSIG_OK;       // (the quirk is a cross Gram's: its X side only)
    if (f32 && (row_end > 0 || row_block_call)) return GPSIG_OK;
    if (c->shard_n > 1) return GPSIG_OK;          // gpsig_set_shard: "entries outside the shard are left untouched" is the pair kernels' contract
    const int M = p->num_levels;
    if (M < 2 || p->order < 1 || p->order > M) return GPSIG_OK;
    const int d = p->num_features * ((raw ? 0 : p->num_lags) + 1);
    SigFeatLaunchFn ffn = sig_feat_lookup(d, M);
    if (!ffn) return GPSIG_OK;
    const bool sym = X2 == nullptr;
    void* out32 = nullptr;                    // float32 calls: where the rounded result goes
    const void* const X_given = X;            // (the caller's pointer: what "sig_features_keep" recognises)
    const int64_t F = sig_feature_count(d, M), ld = (F + 1 + 15) / 16 * 16;
    const int r1 = p->difference ? L1 - 1 : L1, r2 = p->difference ? L2 - 1 : L2;
    if (r1 < 1 || r2 < 1) return GPSIG_OK;
    // One-column state spaces take this route whatever the time model says (round 5): the pair recursion's sums over index tuples cancel by
    // orders of magnitude for scalar increments -- the float64 pair kernels AND the float64 CPU restatement are 2e-3 .. 5e-3 from an 80-bit evaluation on
    // case 779 of the round-5 sweep (tests/golden/fuzz_cases_r5.npz) -- while the per-sequence feature sums do not (8.7e-7 there, float32 rounding).
    if (c->sig_features < 0 && d != 1) {
        // (the higher-order pair kernels carry order^2 grids per level: 34 to 150 times the first order's time at configs[1]'s size)
        const double lattice = double(r1) * r2 * (2.0 * d + 3.0 * M - 1.0) * (p->order > 1 ? 8.0 : 1.0) * (cosine ? 1.5 : 1.0) * (f32 ? 0.5 : 1.0), feat = 2.0 * double(F);      // (the float32 pair kernels run at twice the float64 ones' rate)
        const double pairs = sym ? double(N1) * N1 / 2 : double(N1) * N2;
        if (!(feat * 0.6 < lattice)) return GPSIG_OK;
        // small problems: the contraction is three or four launches with a floor of ~60 us plus its feature kernel (60 us per sequence
        // and CU at 32,768 top-level features), the pair kernels one launch with a floor of ~100 us; rates as measured
        // (tools/bench_crossover.py, profiles/r03_crossover.txt)
        const double seqs = double(N1) + (sym ? 0.0 : double(N2)), t_seq = 60e-6 * double(sig_ipow(d, M)) / 32768.0;
        const double t_feat_kernel = ceil(seqs / 256.0) * t_seq > 15e-6 ? ceil(seqs / 256.0) * t_seq : 15e-6;
        const double t_lattice = 100e-6 + pairs * lattice / (d > 8 ? 12.5e12 : 25e12), t_features = 60e-6 + t_feat_kernel + pairs * feat / 60e12;
        if (pairs < 1024.0 || !(t_features < t_lattice)) return GPSIG_OK;
    }
    const size_t lds = sig_features_lds_bytes(d, M, L1 > L2 ? L1 : L2);
    if (lds > 150 * 1024) return GPSIG_OK;
    // (row_block_call: an EMPTY row block -- a rank that owns no rows -- takes the decisions of a non-empty one and returns before launching)
    const bool rows = row_end > 0 || row_block_call;
    const int64_t NA = rows ? row_end - row_begin : N1, H = N1 / 2;
    int64_t NB = sym ? N1 : N2;
    if (rows) NB = (NA + H < N1) ? NA + H : N1;
    const int nti = int((NA + SG_BM - 1) / SG_BM), ntj = int((NB + SG_BN - 1) / SG_BN);
    const bool symtiles = sym && !rows;
    const int ntiles = symtiles ? nti * (nti + 1) / 2 : nti * ntj;
    const int nslab_all = int((ld + SG_BK - 1) / SG_BK);
    // Workgroups per tile along the depth.  One piece per tile leaves a launch of a few hundred tiles with a nearly empty last round
    // (528 tiles on the chip's 512 workgroup slots: two rounds) and a launch of a few tiles with most of the chip idle; many pieces cost
    // a partial sum each to write and add (the whole result once per piece).  The count minimises a small model of the launch --
    // rounds of 512 workgroups x (slabs per piece x 3.6 us + 10 us; 1.8 us per slab while no CU holds two workgroups) + 2 x pieces x result bytes at 3 TB/s --
    // evaluated for the FULL problem (the Gram's total size and the depth), not for the tiles of this call: the order in which an
    // entry's products are added is then the same whichever tile, row block or rank computes it, and row blocks
    // (gpsig_kernel_K_symm_rows*) reassemble the one-call Gram bit for bit.  16 pieces at N = 4,096 (configs[1]); large Grams keep
    // at least 4 (a rank's chunk of one is a launch of ~1,000 tiles); pieces of at least 8 slabs.
    int nsplit = 1, graded = 1;         // equal depth pieces; the last of them cut into `graded` finer ones
    {
        const int64_t nt_full = (N1 + SG_BM - 1) / SG_BM;
        const int64_t tiles_full = sym ? nt_full * (nt_full + 1) / 2 : nt_full * ((N2 + SG_BN - 1) / SG_BN);
        const double result_bytes = 8.0 * double(N1) * double(sym ? N1 : N2) * (sym ? 0.5 : 1.0);
        double best = 1e300;
        for (int ns = 1; ns <= 128 && ns * 8 <= nslab_all + 7; ++ns) {
            const double wgs = double(tiles_full) * ns, per_piece = double(nslab_all) / ns;
            const double t = (wgs <= 256.0 ? per_piece * 1.8e-6 + 10e-6 : ceil(wgs / 512.0) * (per_piece * 3.6e-6 + 10e-6)) +
                             (ns > 1 ? 2.0 * ns * result_bytes / 3e12 : 0.0);          // (a piece's sum is written, then read)
            if (t < best * 0.999) { best = t; nsplit = ns; }
        }
        if (tiles_full > 1024 && nsplit < 4 && nslab_all >= 32) nsplit = 4;
        // Equal pieces end in a last round of workgroups that takes as long as the others but is only partly full (configs[1]: 528 tiles
        // x 11 pieces = 11.34 rounds of 512, paid as 12).  Cutting the LAST piece into finer ones of halving size (sig_piece_bounds) lets
        // the launch end within one small piece of (total work / slots): the waste of the equal pieces' last round against the finest
        // piece plus a partial sum per extra piece (round 4; option "sig_graded" 0 keeps the equal pieces).
        const double wgs = double(tiles_full) * nsplit, piece_t = double(nslab_all) / nsplit * 3.6e-6;
        if (c->sig_graded != 0 && wgs > 512.0 && nslab_all / nsplit >= 32) {
            const double rounds = wgs / 512.0, waste_equal = (ceil(rounds) - rounds) * piece_t;
            double best_t = waste_equal;
            for (int g = 2; g <= 5; ++g) {
                const double t = piece_t / double(1 << (g - 1)) + (g - 1) * 2.0 * result_bytes / 3e12;
                if (t < best_t * 0.95 && (nslab_all / nsplit) >> (g - 1) >= 8) { best_t = t; graded = g; }
            }
        }
    }
    const size_t part_one = sizeof(double) * size_t(NA) * NB;
    while (nsplit > 1 && part_one * size_t(nsplit - 1 + graded) > (size_t(40) << 30)) { --nsplit; graded = 1; }       // (the exception to the rule above: 40 GiB of partial sums)
    const int npieces = nsplit - 1 + graded;
    const size_t feat_bytes = sizeof(double) * size_t(ld) * (size_t(N1) + (sym ? 0 : size_t(N2)));
    if (feat_bytes + part_one * size_t(npieces) > (size_t(96) << 30)) return GPSIG_OK;
    {
        // what the scratch buffers would have to grow by, against what the device has left: the pair recursion needs no such memory
        // and is the better answer to a full device than an allocation error
        auto grow = [&](int id, size_t bytes) { return bytes > c->buf[id].cap ? bytes + bytes / 8 + 256 : size_t(0); };
        const size_t extra = grow(B_SF0, sizeof(double) * size_t(ld) * N1 + 64) + (sym ? 0 : grow(B_SF1, sizeof(double) * size_t(ld) * N2 + 64)) +
                             grow(B_SF2, part_one * size_t(npieces) + 64);
        // (not for row blocks: the ranks of a sharded evaluation must all take the same route whatever each device has left -- there a
        // full device is an allocation error of that rank, not a silent change of route that the other ranks do not follow)
        if (extra > (size_t(1) << 30) && !c->capturing && !rows) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                size_t held = 0;                // (a buffer that grows is freed first)
                for (int id : {int(B_SF0), int(B_SF1), int(B_SF2)}) held += grow(id, id == B_SF0 ? sizeof(double) * size_t(ld) * N1 + 64 : id == B_SF1 ? (sym ? 0 : sizeof(double) * size_t(ld) * N2 + 64) : part_one * size_t(npieces) + 64) ? c->buf[id].cap : 0;
                if (extra > free_b + held - (free_b + held) / 16) return GPSIG_OK;
            }
        }
    }
    if (rows && NA <= 0) { *done = true; return GPSIG_OK; }        // an empty row block: the route is decided, there is nothing to compute
    const double* w = nullptr;
    if (!raw) CHK(upload_weights(c, p, &w));
    ScaleParams s;
    CHK(scale_params(c, p, !raw, &s));
    void *phi1, *phi2 = nullptr, *part;
    CHK(ensure(c, B_SF0, sizeof(double) * size_t(ld) * N1 + 64, &phi1));
    if (!sym) CHK(ensure(c, B_SF1, sizeof(double) * size_t(ld) * N2 + 64, &phi2));
    CHK(ensure(c, B_SF2, part_one * size_t(npieces) + 64, &part));
    if (f32) {
        void *x64, *y64 = nullptr, *o64;
        const int64_t per1 = int64_t(L1) * p->num_features, per2 = int64_t(L2) * p->num_features;
        CHK(ensure(c, B_SF3, sizeof(double) * size_t(N1 * per1) + 64, &x64));
        HIPCHK(c, sig_convert_launch(X, x64, N1 * per1, true, c->stream));
        if (!sym) {
            CHK(ensure(c, B_SF4, sizeof(double) * size_t(N2 * per2) + 64, &y64));
            HIPCHK(c, sig_convert_launch(X2, y64, N2 * per2, true, c->stream));
        }
        CHK(ensure(c, B_SF5, sizeof(double) * size_t(return_levels ? M + 1 : 1) * size_t(N1) * size_t(sym ? N1 : N2) + 64, &o64));
        out32 = out; X = x64; X2 = sym ? nullptr : y64; out = o64;
    }
    const int normalize = (!raw && p->normalization) ? 1 : 0;
    auto features = [&](const void* Xs, int64_t N, int L, void* phi, bool squared) -> int {
        SigFeatArgs A;
        memset(&A, 0, sizeof(A));
        A.X = static_cast<const double*>(Xs); A.N = N; A.L = L; A.difference = p->difference ? 1 : 0; A.P = s;
        A.w = w; A.normalize = normalize; A.jitter = p->jitter; A.Phi = static_cast<double*>(phi); A.ld = ld; A.dlev = nullptr;
        A.order = p->order;
        A.unit_points = cosine ? 1 : 0;
        A.norm_squared = squared ? 1 : 0;
        const int64_t cap = sig_threads(d, M) <= 128 ? 16384 : 4096;      // (one- or two-wavefront workgroups are latency-bound at 16 per CU)
        const unsigned grid = unsigned(N < cap ? N : cap);
        hipError_t e = ffn(A, grid, sig_features_lds_bytes(d, M, L), c->stream);
        if (e != hipSuccess) return fail(c, GPSIG_ERR_HIP, "sig_features_kernel: %s", hipGetErrorString(e));
        return GPSIG_OK;
    };
    // "sig_features_keep": between the calls of ONE decomposed evaluation (the chunks of a rank's row block: parallel.ShardedGram) the
    // features of the same sequences under the same parameters are built once.  The caller promises not to write X in between.
    uint64_t key = 1469598103934665603ull;
    auto mix = [&](const void* ptr, size_t bytes) {
        const unsigned char* b = static_cast<const unsigned char*>(ptr);
        for (size_t i = 0; i < bytes; ++i) { key ^= b[i]; key *= 1099511628211ull; }
    };
    {
        const int32_t head[8] = {p->base_kernel, p->dtype, p->num_features, p->num_levels, p->order, p->difference, p->normalization, p->num_lags};
        mix(head, sizeof(head));
        mix(&p->sigma, sizeof(double)); mix(&p->jitter, sizeof(double));
        mix(p->variances, sizeof(double) * (M + 1));
        if (p->lengthscales) mix(p->lengthscales, sizeof(double) * p->num_features);
        if (p->num_lags > 0) { mix(p->lags, sizeof(double) * p->num_lags); mix(p->gamma, sizeof(double) * (p->num_lags + 1)); }
        const int64_t tail[4] = {N1, L1, (raw ? 1 : 0) + (x_squared ? 2 : 0), ld};
        mix(tail, sizeof(tail));
    }
    const bool reuse = c->sf_keep && c->sf_valid && c->sf_X == X_given && c->sf_phi == phi1 && c->sf_key == key;
    if (N1 > 0 && !reuse) CHK(features(X, N1, L1, phi1, x_squared != 0));
    c->sf_valid = c->sf_keep != 0 && N1 > 0;
    c->sf_X = X_given; c->sf_phi = phi1; c->sf_key = key;
    if (!sym && N2 > 0) CHK(features(X2, N2, L2, phi2, false));
    if (NA <= 0 || NB <= 0) { *done = true; return GPSIG_OK; }
    // level sums of the weights: the exact diagonal of the normalised symmetric Gram (kernels.py:430-433: (K_ii + jitter) / (K_ii + jitter))
    const int nlev = return_levels ? M + 1 : 1;
    const int64_t Ncols = sym ? N1 : N2;
    for (int lv = 0; lv < nlev; ++lv) {
        int kb = 0, ke = int(F) + 1;                         // all levels and the level-0 column: the level sum is inside the contraction
        double dval = 0.0;
        if (return_levels) {
            if (lv == 0) { kb = int(F); ke = int(F) + 1; }
            else { kb = sig_feature_count(d, lv - 1); ke = sig_feature_count(d, lv); }
            dval = raw ? 0.0 : p->sigma * p->variances[lv];
        } else {
            for (int m = 0; m <= M; ++m) dval += p->sigma * p->variances[m];
        }
        SigGramArgs G;
        memset(&G, 0, sizeof(G));
        G.A = static_cast<const double*>(phi1) + (rows ? row_begin * ld : 0);
        G.B = static_cast<const double*>(sym ? phi1 : phi2);
        G.NA = NA; G.NB = NB; G.lda = ld; G.ldb = ld;
        G.b_off = rows ? ((row_begin - H) % N1 + N1) % N1 : 0; G.b_mod = sym ? N1 : N2;
        G.k_begin = kb; G.k_end = ke;
        G.band = (rows && NA + H <= N1) ? H : 0;      // column c of the block is sequence row_begin - H + c (no wrap inside the block): row i owns c in [i, i + H]
        const int nslab = (ke - kb + SG_BK - 1) / SG_BK;
        // (the per-level products of return_levels: equal pieces of their own, shorter depth)
        const int ns = sig_piece_bounds(nslab, nsplit, return_levels ? 1 : graded, G.bound);
        G.nsplit = ns; G.symmetric = symtiles ? 1 : 0; G.ntj = ntj; G.part = static_cast<double*>(part);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        bool on = false;
        if (timed) CHK(timing_begin_any(c, &e0, &e1, &on));
        // the LDS-DMA form reads whole slabs: only where the columns behind k_end are the zero padding of the feature rows
        int used_dma = 0;
        HIPCHK(c, sig_gram_launch(G, ntiles, c->stream, (c->sig_gemm_dma && !return_levels) ? 1 : 0, &used_dma));
        if (on) {
            HIPCHK(c, hipEventRecord(e1, c->stream));
            c->t_launches += 1;
            c->t_pairs += NA * NB;
            c->t_kernel = used_dma ? "sig_gram_dma_kernel" : "sig_gram_kernel";
            c->t_flops += 2.0 * double(ntiles) * SG_BM * SG_BN * double(nslab) * SG_BK;
        }
        SigReduceArgs R;
        memset(&R, 0, sizeof(R));
        R.part = static_cast<const double*>(part); R.nsplit = ns; R.NA = NA; R.NB = NB;
        R.out = static_cast<double*>(out) + (return_levels ? int64_t(lv) * N1 * Ncols : 0);
        R.so_i = Ncols; R.so_j = 1;
        R.mode = rows ? 2 : (symtiles ? 1 : 0);
        R.diag_set = (sym && normalize) ? 1 : 0; R.diag_value = dval;
        R.N = N1; R.r0 = row_begin; R.c0 = G.b_off; R.compact = compact;
        HIPCHK(c, sig_reduce_launch(R, c->stream));
    }
    if (out32) HIPCHK(c, sig_convert_launch(out, out32, int64_t(return_levels ? M + 1 : 1) * N1 * (sym ? N1 : N2), false, c->stream));
    *done = true;
    return GPSIG_OK;
}

template <typename TT>
struct Impl {
// ---- seq-gram planning -----------------------------------------------------------------------------
struct SeqPlanned {
    SeqConfig cfg;
    SeqLaunchFn fn;
    int mode, d_eff;
    bool rbf_prescaled;      // fn is an RBF (or, float64, Matern) instance that takes prescaled records: points x prescale, -|row|^2/2 in the spare column
    int fast_kind;           // the base kernel fn has at compile time on prescaled records (float64: BASE_RBF or a Matern family), else -1
    double prescale;         // EXP_PRESCALE (float64, table-driven exp) or PK2_RBF_PRESCALE (float32, v_exp_f32)
    bool pk2;                // fn is a seq_pk2_kernel instance:
    int ny, waves;           //   a pair group serves ny y sequences, a workgroup has `waves` wavefronts on one x ring
};

// pairs_hint: how many pairs the launch this plan is for will evaluate (0: unknown / small)
static int plan_seq(gpsig_ctx* c, const gpsig_params* p, int d_eff, int Ly, SeqPlanned* out, int64_t pairs_hint = 0) {
    SeqGeom g0 = seq_geometry(p->base_kernel, p->difference, Ly, 4, int(sizeof(TT)));
    out->rbf_prescaled = false;
    out->fast_kind = -1;
    out->prescale = 1.0;
    out->pk2 = false;
    out->ny = 1; out->waves = 1;
    if (p->base_kernel == GPSIG_BASE_SPECTRAL) {
        // takes the points, not inner products: wavefront kernels with this family at compile time (seq_step_spectral) for float64,
        // first order, with differences, d <= 16; everything else through the one-pair-per-thread kernel (the callers' fallback)
        int k = -1;
        if (sizeof(TT) == 8 && c->spectral_wave != 0 && g0.mode == MODE_PT_DIFF && !(p->order > 1 && p->num_levels > 1))
            k = seq_select(SEQ_TABLE_SPECTRAL, N_SEQ_TABLE_SPECTRAL, g0.rows, d_eff, p->num_levels, false);
        if (k < 0) return fail(c, GPSIG_ERR_UNSUPPORTED, "the spectral base kernel: no wavefront kernel for this shape / dtype / order");
        out->cfg = SEQ_TABLE_SPECTRAL[k];
        out->mode = g0.mode;
        out->d_eff = d_eff;
        out->fn = out->cfg.G == 16 ? seq_lookup_ptd_spectral_g16(out->cfg.G, out->cfg.C, out->cfg.D, out->cfg.MMAX, false)
                                   : seq_lookup_ptd_spectral_g64(out->cfg.G, out->cfg.C, out->cfg.D, out->cfg.MMAX, false);
        if (!out->fn) return fail(c, GPSIG_ERR_UNSUPPORTED, "spectral seq-gram kernel shape missing from this build");
        return GPSIG_OK;
    }
    if (p->order > 1 && p->num_levels > 1) {            // higher-order algorithm (signature_algs.py:37-74)
        int k = seq_select_ho(SEQ_HO_TABLE, N_SEQ_HO_TABLE, g0.rows, d_eff, p->num_levels, p->order);
        if (k < 0)
            return fail(c, GPSIG_ERR_UNSUPPORTED,
                        "no higher-order seq-gram kernel shape for %d record rows, d=%d, num_levels=%d, order=%d (built: order <= 8 up "
                        "to 64 rows, <= 4 up to 128 rows, 2 up to 512 rows; num_levels <= 6 (8 up to 64 rows); d <= 16, or d <= 32 with at most 128 rows)",
                        g0.rows, d_eff, p->num_levels, p->order);
        const SeqHOConfig& h = SEQ_HO_TABLE[k];
        out->cfg = SeqConfig{h.G, h.C, h.D, h.MMAX, false};
        out->mode = g0.mode;
        out->d_eff = d_eff;
        out->fn = seq_launcher_ho(g0.mode, h, sizeof(TT) == 4);
        // round 6: exact instances (num_levels and order at compile time, the RBF kernel on prescaled records with the table exp)
        if (sizeof(TT) == 8 && c->allow_exact && g0.mode == MODE_PT_DIFF && p->base_kernel == GPSIG_BASE_RBF) {
            SeqLaunchFn ex = seq_lookup_ho_ptdrbf_exact(h.G, h.C, h.D, p->num_levels, p->order);
            if (!ex) ex = seq_lookup_ho_ptdrbf_exact_o4(h.G, h.C, h.D, p->num_levels, p->order);
            if (ex) {
                out->fn = ex;
                out->cfg = SeqConfig{h.G, h.C, h.D, p->num_levels, true};
                out->rbf_prescaled = true;               // (fast_kind stays -1: the stash instances are first-order)
                out->prescale = SEQ_RBF_PRESCALE;
            }
        }
        if (sizeof(TT) == 8 && c->allow_exact && g0.mode == MODE_PT_DIFF && seq_is_matern(p->base_kernel) && c->matern_fast != 0) {
            SeqLaunchFn ex = p->base_kernel == GPSIG_BASE_MATERN12 ? seq_lookup_ho_ptdm12_exact(BASE_MATERN12, h.G, h.C, h.D, p->num_levels, p->order)
                             : (p->base_kernel == GPSIG_BASE_MATERN32 ? seq_lookup_ho_ptdm32_exact(BASE_MATERN32, h.G, h.C, h.D, p->num_levels, p->order)
                                                                      : seq_lookup_ho_ptdm52_exact(BASE_MATERN52, h.G, h.C, h.D, p->num_levels, p->order));
            if (ex) {
                out->fn = ex;
                out->cfg = SeqConfig{h.G, h.C, h.D, p->num_levels, true};
                out->rbf_prescaled = true;               // (prescaled records; the spare column is written and ignored)
                out->prescale = seq_matern_prescale(p->base_kernel);
            }
        }
        if (!out->fn) return fail(c, GPSIG_ERR_UNSUPPORTED, "higher-order kernel shape missing from this build");
        return GPSIG_OK;
    }
    const bool f32 = sizeof(TT) == 4;
    // float32, first order, exact num_levels: seq_pk2_kernel -- two y sequences per pair group on the packed instructions, and for
    // launches large enough to fill the chip with 4-wave workgroups, four wavefronts on one x ring (BASELINE configs[4], RBF / linear:
    // 41.6 / 28.0 ms one-sequence kernels, 38.6 / 33.5 ms packed, 30.5 / 26.2 ms packed + shared ring; profiles/r02_bench_c5_variants.txt).
    // The linear kernel gains only with the shared ring, so small launches keep the one-sequence kernels for it.
    if (f32 && c->allow_pk2 && c->allow_exact && (g0.mode == MODE_INC || (g0.mode == MODE_PT_DIFF && p->base_kernel == GPSIG_BASE_RBF))) {
        int G, C, D;
        const bool big = pairs_hint >= (int64_t(1) << 18);
        const int waves = c->f32_waves > 0 ? c->f32_waves : (big ? 4 : 1);
        const bool take = g0.mode == MODE_PT_DIFF || waves == 4 || c->allow_pk2 == 2;
        if (take && seq_pk2_select(g0.rows, d_eff, p->num_levels, &G, &C, &D)) {
            out->cfg = SeqConfig{G, C, D, p->num_levels, true};
            out->mode = g0.mode;
            out->d_eff = d_eff;
            out->ny = c->f32_pack == 1 ? 1 : 2;
            out->waves = waves == 4 ? 4 : 1;
            out->fn = seq_pk2_lookup(G, C, D, p->num_levels, g0.mode, out->ny, out->waves);
            out->pk2 = true;
            out->rbf_prescaled = g0.mode == MODE_PT_DIFF;
            out->prescale = PK2_RBF_PRESCALE;
            if (out->fn) return GPSIG_OK;
        }
    }
    out->pk2 = false; out->rbf_prescaled = false; out->prescale = 1.0; out->ny = 1; out->waves = 1;
    const SeqConfig* tab = f32 ? SEQ_TABLE_F32 : (g0.mode == MODE_PT_NODIFF ? SEQ_TABLE_GENERIC : SEQ_TABLE);
    const int ntab = f32 ? N_SEQ_TABLE_F32 : (g0.mode == MODE_PT_NODIFF ? N_SEQ_TABLE_GENERIC : N_SEQ_TABLE);
    int k = seq_select(tab, ntab, g0.rows, d_eff, p->num_levels, c->allow_exact != 0 && g0.mode != MODE_PT_NODIFF);
    if (k < 0)
        return fail(c, GPSIG_ERR_UNSUPPORTED,
                    "no seq-gram kernel shape for %d record rows on the register-resident side, d=%d, num_levels=%d "
                    "(built: rows <= 512 for d*(num_lags+1) <= 8, <= 256 up to 16, <= 128 up to 32; num_levels <= 8)",
                    g0.rows, d_eff, p->num_levels);
    out->cfg = tab[k];
    out->mode = g0.mode;
    out->d_eff = d_eff;
    out->fn = nullptr;
    if (!f32 && g0.mode == MODE_PT_DIFF && p->base_kernel == GPSIG_BASE_RBF && tab[k].exact) {
        out->fn = seq_launcher_rbf(tab[k]);
        out->rbf_prescaled = out->fn != nullptr;
        out->prescale = SEQ_RBF_PRESCALE;
        if (out->fn) out->fast_kind = BASE_RBF;
    }
    // the Matern families at compile time on prescaled records (round 5: seq_step_matern_prescaled), the shapes of GPSIG_SEQ_CONFIGS_EXACT
    if (!f32 && g0.mode == MODE_PT_DIFF && seq_is_matern(p->base_kernel) && tab[k].exact && c->matern_fast != 0) {
        const SeqConfig& t = tab[k];
        out->fn = p->base_kernel == GPSIG_BASE_MATERN12 ? seq_lookup_ptdm12_exact(t.G, t.C, t.D, t.MMAX, t.exact)
                  : (p->base_kernel == GPSIG_BASE_MATERN32 ? seq_lookup_ptdm32_exact(t.G, t.C, t.D, t.MMAX, t.exact)
                                                           : seq_lookup_ptdm52_exact(t.G, t.C, t.D, t.MMAX, t.exact));
        out->rbf_prescaled = out->fn != nullptr;            // (the records' spare column is written and ignored)
        out->prescale = seq_matern_prescale(p->base_kernel);
        if (out->fn) out->fast_kind = p->base_kernel;
    }
    if (!out->fn) out->fn = seq_launcher(g0.mode, tab[k], sizeof(TT) == 4, p->base_kernel);
    if (!out->fn) return fail(c, GPSIG_ERR_UNSUPPORTED, "seq-gram kernel shape missing from this build");
    return GPSIG_OK;
}

// records of N sequences (device, user layout (N, L, d)) into buffer `id`
static int make_records(gpsig_ctx* c, const gpsig_params* p, bool apply_scaling, const SeqPlanned& pl, const void* Xdev, int64_t N,
                 int L, int id, const void** rec, SeqGeom* geom) {
    *geom = seq_geometry(p->base_kernel, p->difference, L, pl.cfg.D, int(sizeof(TT)));
    const size_t bytes = size_t(N) * geom->rec_elems * sizeof(TT);
    void* d;
    CHK(ensure(c, id, bytes ? bytes : 8, &d));
    if (N > 0) {
        CHK(zero_async(c, d, bytes));
        ScaleParams s;
    CHK(scale_params(c, p, apply_scaling, &s));
        const int64_t total = N * geom->rows * s.d_eff();
        hipLaunchKernelGGL(prep_seq_records_kernel<TT>, dim3(grid_for(total)), dim3(256), 0, c->stream,
                           static_cast<const TT*>(Xdev), N, L, s, geom->mode, p->difference, geom->rows, geom->RS,
                           int64_t(geom->rec_elems), static_cast<TT*>(d), pl.rbf_prescaled ? TT(pl.prescale) : TT(1),
                           pl.rbf_prescaled ? pl.cfg.D : -1);
        HIPCHK(c, hipGetLastError());
    }
    *rec = d;
    return GPSIG_OK;
}

// Timing covers the launches since gpsig_timing_reset, up to TIMING_MAX_EVENTS of them (a long-running caller that never
// reads the timing must not accumulate events); *on says whether this launch is timed.  Never inside a graph capture.
static int timing_begin(gpsig_ctx* c, hipEvent_t* e0, hipEvent_t* e1, bool* on) { return timing_begin_any(c, e0, e1, on); }

struct SeqRun {
    const void* xrec; const void* yrec;
    SeqGeom gx, gy;
    int64_t N1, N2;
    void* out; int64_t si, sj, sm;
    const void* ax; const void* by;
    double jitter_diag;
    int sum_levels, pred, mirror;
    bool timed;
    int64_t y_begin, y_end;   // y-block range (0, 0) = all
    int compact;              // SeqGramArgs::compact
};

static int launch_seq(gpsig_ctx* c, const gpsig_params* p, const SeqPlanned& pl, const SeqRun& r) {
    if (r.N1 <= 0 || r.N2 <= 0) return GPSIG_OK;
    if (r.N1 > 0x7fffffff || r.N2 > 0x7fffffff) return fail(c, GPSIG_ERR_UNSUPPORTED, "more than 2^31 sequences");
    const int ypb = (64 / pl.cfg.G) * pl.ny * pl.waves;
    // aim for ~64k independent tasks (about 20 per resident wave slot) so the tail is a few per cent
    const int64_t nblocks = ((r.y_end > 0 ? r.y_end - r.y_begin : r.N2) + ypb - 1) / ypb;
    // diagonal with several pair groups per wavefront: each group sweeps its own sequence (SeqGramArgs::diag_own)
    const bool diag_own = r.pred == PRED_DIAG && !pl.pk2 && ypb > 1 && ypb == 64 / pl.cfg.G && c->diag_own != 0;
    const int64_t xtot = r.pred == PRED_ALL ? r.N1 : (r.pred == PRED_DIAG ? (diag_own ? 1 : ypb) : r.N1 / 2 + ypb);
    // ... unless the whole problem is smaller than that: then short runs, so that a small evaluation is spread over the chip
    // instead of a few wavefronts sweeping eight pairs in a row
    int64_t max_run = (xtot * nblocks + 65535) / 65536;
    const int64_t run_floor = xtot * nblocks >= 8 * 4096 ? 8 : (xtot * nblocks / 4096 > 1 ? xtot * nblocks / 4096 : 1);
    if (max_run < run_floor) max_run = run_floor;
    if (max_run > 256) max_run = 256;
    if (c->max_run > 0) max_run = c->max_run;
    // the task list is a function of these integers only: reuse the device copy while they stay the same
    const int64_t key[10] = {r.N1, r.N2, ypb, r.pred, max_run, c->shard_i, c->shard_n, r.y_begin, r.y_end > 0 ? r.y_end : -1, diag_own ? 2 : 1};
    const SeqTask* dt = nullptr;
    int ntasks = 0;
    int64_t npairs = 0;
    CHK(task_list(c, key, [&](std::vector<SeqTask>& T) {
        T = seq_build_tasks(r.N1, r.N2, ypb, r.pred, diag_own ? ypb : int(max_run), c->shard_i, c->shard_n, r.y_begin, r.y_end > 0 ? r.y_end : -1);
        if (diag_own)
            for (SeqTask& t : T) t.nx = 1;               // (y0, x0 = y0): one sweep, group g against sequence y0 + g
        int64_t pairs = 0;
        for (const SeqTask& t : T) pairs += int64_t(t.nx) * ypb;
        return pairs;
    }, &dt, &ntasks, &npairs));
    if (ntasks == 0) return GPSIG_OK;

    SeqGramArgs A;
    memset(&A, 0, sizeof(A));
    A.xrec = r.xrec; A.yrec = r.yrec; A.tasks = dt;
    A.N1 = r.N1; A.N2 = r.N2;
    A.xrec_stride = r.gx.rec_elems; A.yrec_stride = r.gy.rec_elems;
    A.R1 = r.gx.rows; A.R2 = r.gy.rows; A.RS = r.gx.RS; A.M = p->num_levels; A.order = p->order;
    A.nslot = seq_ring(pl.cfg.G, r.gx.rows).nslot;
    if (diag_own && A.nslot < ypb) A.nslot = ypb;
    A.issue_at = seq_ring(pl.cfg.G, r.gx.rows).issue_at;
    A.slot_elems = r.gx.rec_elems;
    A.kind = p->base_kernel;
    base_p(p, &A.p0, &A.p1);
    if (p->base_kernel == GPSIG_BASE_SPECTRAL) CHK(spectral_table(c, p, &A.spec));
    A.out = r.out; A.si = r.si; A.sj = r.sj; A.sm = r.sm;
    A.ax = r.ax; A.by = r.by; A.jitter_diag = r.jitter_diag;
    A.sum_levels = r.sum_levels; A.pred = diag_own ? int(PRED_DIAG_OWN) : r.pred; A.mirror = r.mirror; A.use_glds = c->use_glds; A.compact = r.compact; A.keep_reset = c->keep_reset;
    const size_t lds = sizeof(TT) * (size_t(A.RS) + size_t(A.nslot) * A.slot_elems);
    if (lds > 160 * 1024) return fail(c, GPSIG_ERR_UNSUPPORTED, "x-side records of %d rows do not fit the LDS ring (%zu bytes)", A.R1, lds);
    // gpsig_seq_gram_levels_stash: the instances the fused reverse kernel continues from also write what it needs of this recursion
    SeqLaunchFn fn = pl.fn;
    const bool st_rbf = pl.fast_kind == BASE_RBF, st_matern = pl.fast_kind >= 0 && seq_is_matern(pl.fast_kind);
    if (c->stash_want && sizeof(TT) == 8 && !pl.pk2 && (st_rbf || st_matern) && pl.cfg.exact && pl.cfg.G == 16 && pl.cfg.C == 4 && p->order <= 1 &&
        (r.pred == PRED_ALL || r.pred == PRED_CIRCULANT) && c->shard_n == 1 && r.y_begin == 0 && r.y_end <= 0 && !r.compact && r.gx.rows >= 2 &&
        r.gx.rows <= 64 && r.gy.rows <= 64) {
        const int mm = p->num_levels == pl.cfg.MMAX ? pl.cfg.MMAX : -1;
        SeqLaunchFn sfn = st_rbf ? seq_lookup_ptdrbf_stash(16, 4, pl.cfg.D, mm) : seq_lookup_ptdmatern_stash(pl.fast_kind, 16, 4, pl.cfg.D, mm);
        const int R1l = r.gx.rows - 1, LQ = p->num_levels - 1;
        const int64_t stride = int64_t(R1l) * LQ + 16 * int64_t(LQ) * 4;        // grad_fused_kernel.hpp: fused_stash_stride
        const size_t need = sizeof(double) * size_t(npairs) * size_t(stride);
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(c->stream, &cs);
        // (kept only where the memory is there: a buffer that has to grow must leave a gibibyte free -- the stash is an optimisation, an
        // evaluation never fails for want of it)
        bool room = c->buf[B_STASH].cap >= need + 64;
        if (!room) {
            size_t free_b = 0, total_b = 0;
            room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + c->buf[B_STASH].cap > need + (size_t(1) << 30);
        }
        if (sfn && LQ >= 1 && room && cs == hipStreamCaptureStatusNone && need <= (size_t(c->grad_stash_mb > 0 ? c->grad_stash_mb : 0) << 20)) {
            void* st;
            CHK(ensure(c, B_STASH, need + 64, &st));
            int64_t k2[10];
            for (int q = 0; q < 10; ++q) k2[q] = key[q];
            k2[9] = 4 + (diag_own ? 0 : 0);                                     // the tasks' first pair slots, packed (y0 = low word, x0 = high word)
            const SeqTask* dp = nullptr;
            int np = 0;
            CHK(task_list(c, k2, [&](std::vector<SeqTask>& T) {
                std::vector<SeqTask> F = seq_build_tasks(r.N1, r.N2, ypb, r.pred, int(max_run), 0, 1);
                T.clear();
                int64_t at = 0;
                for (const SeqTask& t : F) { T.push_back(SeqTask{int32_t(uint32_t(at & 0xffffffff)), int32_t(at >> 32), 0}); at += t.nx; }
                return at;
            }, &dp, &np));
            if (np == ntasks) {
                A.stash = static_cast<double*>(st); A.stash_pair0 = dp; A.stash_stride = stride;
                fn = sfn;
                c->stash_desc[0] = c->stash_gen = next_stash_generation(); c->stash_desc[1] = r.pred; c->stash_desc[2] = max_run; c->stash_desc[3] = ypb;
                c->stash_desc[4] = ntasks; c->stash_desc[5] = npairs; c->stash_desc[6] = stride; c->stash_desc[7] = R1l;
            }
        }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool timed = false;
    if (r.timed) CHK(timing_begin(c, &e0, &e1, &timed));
    HIPCHK(c, fn(A, ntasks, lds, c->stream));
    if (timed) {
        HIPCHK(c, hipEventRecord(e1, c->stream));
        c->t_launches += 1;
        c->t_pairs += npairs;
    }
    return GPSIG_OK;
}

// diag levels of N sequences, sequence-major (N, M+1), into buffer `id`
static int diag_levels(gpsig_ctx* c, const gpsig_params* p, const SeqPlanned& pl, const void* rec, const SeqGeom& g, int64_t N,
                int id, const void** dlev) {
    const int M1 = p->num_levels + 1;
    void* d;
    CHK(ensure(c, id, sizeof(TT) * size_t(N) * M1 + 8, &d));
    SeqRun r;
    memset(&r, 0, sizeof(r));
    r.xrec = rec; r.yrec = rec; r.gx = g; r.gy = g; r.N1 = N; r.N2 = N;
    r.out = d; r.si = M1; r.sj = 0; r.sm = 1;
    r.sum_levels = 0; r.pred = PRED_DIAG; r.mirror = 0; r.timed = false;
    // the diag pass must cover every sequence on every shard: factors are needed for all rows/columns
    const int si = c->shard_i, sn = c->shard_n;
    c->shard_i = 0; c->shard_n = 1;
    int rc = launch_seq(c, p, pl, r);
    c->shard_i = si; c->shard_n = sn;
    CHK(rc);
    *dlev = d;
    return GPSIG_OK;
}

static int make_factors(gpsig_ctx* c, const void* dlev, int64_t N, int M1, const double* w, double jitter, int id, const void** fac,
                 int squared = 0) {
    void* d;
    CHK(ensure(c, id, sizeof(TT) * size_t(N) * M1 + 8, &d));
    if (N > 0) {
        hipLaunchKernelGGL(factors_kernel<TT>, dim3(grid_for(N * M1)), dim3(256), 0, c->stream,
                           static_cast<const TT*>(dlev), N, M1, w, jitter, squared, static_cast<TT*>(d));
        HIPCHK(c, hipGetLastError());
    }
    *fac = d;
    return GPSIG_OK;
}

// ---- any-shape fallback (seq_levels_generic_kernel): float64, first-order algorithm --------------------------------------
static bool generic_ok(const gpsig_params* p) { return sizeof(TT) == 8 && p->num_levels <= 8 && p->order <= 8; }

// raw levels of the pairs (i, j) (or (i, i) when diag) into out[m * sm + i * si + j * sj]
static int generic_levels(gpsig_ctx* c, const gpsig_params* p, bool apply_scaling, const void* X, const void* Y, int64_t N1, int64_t N2,
                          int L1, int L2, bool diag, double* out, int64_t sm, int64_t si, int64_t sj) {
    if (N1 == 0 || N2 == 0) return GPSIG_OK;
    ScaleParams sp;
    CHK(scale_params(c, p, apply_scaling, &sp));
    const int d_eff = sp.d_eff(), M = p->num_levels;
    // wide route (wide_api.hip): the argument lattices by dgemm, one wavefront per lattice -- the distance kernels at order 1, up to 512 lattice columns
    if (wide_lat_available(c, p, L1, L2) && sm == (diag ? N1 : N1 * N2) && si == (diag ? 1 : N2) && sj == (diag ? 0 : 1)) {
        const bool same_ = diag || Y == nullptr || Y == X;
        void *xs, *ys = nullptr;
        CHK(ensure(c, B_GR0, sizeof(double) * size_t(N1) * L1 * d_eff + 8, &xs));
        hipLaunchKernelGGL(prep_seq_scaled_kernel<double>, dim3(grid_for(N1 * int64_t(L1) * d_eff)), dim3(256), 0, c->stream, static_cast<const double*>(X), N1,
                           L1, sp, static_cast<double*>(xs));
        HIPCHK(c, hipGetLastError());
        if (!same_) {
            CHK(ensure(c, B_GR1, sizeof(double) * size_t(N2) * L2 * d_eff + 8, &ys));
            hipLaunchKernelGGL(prep_seq_scaled_kernel<double>, dim3(grid_for(N2 * int64_t(L2) * d_eff)), dim3(256), 0, c->stream, static_cast<const double*>(Y),
                               N2, L2, sp, static_cast<double*>(ys));
            HIPCHK(c, hipGetLastError());
        }
        return wide_lat_forward(c, p, d_eff, static_cast<const double*>(xs), static_cast<const double*>(ys), N1, N2, L1, L2, diag, out);
    }
    const int64_t s1 = (N1 + 63) / 64 * 64, s2 = (N2 + 63) / 64 * 64;
    void *xt, *yt = nullptr;
    CHK(ensure(c, B_GR0, sizeof(double) * size_t(L1) * d_eff * s1 + 8, &xt));
    hipLaunchKernelGGL(prep_seq_timemajor_kernel<double>, dim3(grid_for(int64_t(L1) * d_eff * s1)), dim3(256), 0, c->stream,
                       static_cast<const double*>(X), N1, s1, L1, sp, static_cast<double*>(xt));
    HIPCHK(c, hipGetLastError());
    const bool same = diag || Y == nullptr || Y == X;
    if (!same) {
        CHK(ensure(c, B_GR1, sizeof(double) * size_t(L2) * d_eff * s2 + 8, &yt));
        hipLaunchKernelGGL(prep_seq_timemajor_kernel<double>, dim3(grid_for(int64_t(L2) * d_eff * s2)), dim3(256), 0, c->stream,
                           static_cast<const double*>(Y), N2, s2, L2, sp, static_cast<double*>(yt));
        HIPCHK(c, hipGetLastError());
    }
    const SeqGeom g = seq_geometry(p->base_kernel, p->difference, L2, 4, 8);
    // the linear kernel without differences is a point kernel here (kappa = <x, y>)
    const int mode = (g.mode == MODE_INC && !p->difference) ? MODE_PT_NODIFF : g.mode;
    const int dr = mode == MODE_PT_NODIFF ? 0 : 1;
    const int R2 = L2 - dr;
    const bool ho = p->order > 1 && M > 1;
    const size_t per_j = sizeof(double) * size_t(M > 1 ? M - 1 : 1) * (ho ? 9 : 1) * size_t(R2 > 0 ? R2 : 1) * size_t(s1);
    int64_t chunk = diag ? 1 : int64_t((size_t(c->grad_scratch_mb > 0 ? c->grad_scratch_mb : 4096) << 20) / per_j);
    if (chunk < 1) chunk = 1;
    if (chunk > N2) chunk = N2;
    if (chunk > 65535) chunk = 65535;
    void* scr;
    CHK(ensure(c, B_GR4, per_j * size_t(chunk) + 64, &scr));
    GenericSeqArgs A;
    memset(&A, 0, sizeof(A));
    A.XT = static_cast<const double*>(xt); A.YT = same ? A.XT : static_cast<const double*>(yt);
    A.xstride = s1; A.ystride = same ? s1 : s2;
    A.N1 = N1; A.N2 = N2; A.L1 = L1; A.L2 = L2; A.d = d_eff; A.M = M; A.kind = p->base_kernel; A.mode = mode; A.diag = diag ? 1 : 0;
    base_p(p, &A.p0, &A.p1);
    CHK(spectral_table(c, p, &A.spec));
    A.scratch = static_cast<double*>(scr);
    A.out = out; A.sm = sm; A.si = si; A.sj = sj;
    for (int64_t j0 = 0; j0 < (diag ? 1 : N2); j0 += chunk) {
        const int64_t nj = diag ? 1 : ((N2 - j0 < chunk) ? N2 - j0 : chunk);
        A.j0 = j0; A.pairs = s1 * nj;
        if (ho) hipLaunchKernelGGL(seq_levels_generic_ho_kernel, dim3(unsigned(s1 / 64), unsigned(nj)), dim3(64), 0, c->stream, A, int(p->order));
        else hipLaunchKernelGGL(seq_levels_generic_kernel, dim3(unsigned(s1 / 64), unsigned(nj)), dim3(64), 0, c->stream, A);
        HIPCHK(c, hipGetLastError());
    }
    return GPSIG_OK;
}

// K / _K_seq for shapes the wavefront kernel is not built for
static int seq_K_generic(gpsig_ctx* c, const gpsig_params* p, bool raw, const void* X, const void* X2, int64_t N1, int64_t N2, int L1, int L2,
                         int return_levels, void* out, int x_squared) {
    const bool sym = X2 == nullptr;
    const int M1 = p->num_levels + 1;
    const int64_t Nc = sym ? N1 : N2;
    const void *fa = nullptr, *fb = nullptr;
    double jitter_diag = 0.0;
    if (!raw) {
        const double* w;
        CHK(upload_weights(c, p, &w));
        if (p->normalization) {
            CHK(side_factors(c, p, true, X, N1, L1, w, x_squared, B_DLEV0, B_FAC0, &fa));
            if (sym) CHK(make_factors(c, c->buf[B_DLEV0].p, N1, M1, nullptr, p->jitter, B_FAC1, &fb));
            else CHK(side_factors(c, p, true, X2, N2, L2, nullptr, 0, B_DLEV1, B_FAC1, &fb));
            if (sym) jitter_diag = p->jitter;
        } else {
            CHK(make_factors(c, nullptr, N1, M1, w, 0.0, B_FAC0, &fa));
        }
    }
    if (raw) return generic_levels(c, p, false, X, sym ? X : X2, N1, Nc, L1, sym ? L1 : L2, false, static_cast<double*>(out), N1 * Nc, Nc, 1);
    void* lev;
    CHK(ensure(c, B_GR5, sizeof(double) * size_t(M1) * N1 * Nc + 8, &lev));
    CHK(generic_levels(c, p, true, X, sym ? X : X2, N1, Nc, L1, sym ? L1 : L2, false, static_cast<double*>(lev), N1 * Nc, Nc, 1));
    if (N1 > 0 && Nc > 0) {
        hipLaunchKernelGGL(levels_epilogue_kernel, dim3(grid_for(N1 * Nc)), dim3(256), 0, c->stream, static_cast<const double*>(lev), N1, Nc, M1,
                           static_cast<const double*>(fa), static_cast<const double*>(fb), jitter_diag, return_levels ? 0 : 1,
                           static_cast<double*>(out));
        HIPCHK(c, hipGetLastError());
    }
    return GPSIG_OK;
}

// Per-sequence factors w[m] / sqrt(diag_m + jitter) (or squared) of one side, with that side's own kernel shape:
// the diagonal pass keeps the sequence itself in registers, whatever the main pass does with it.
static int side_factors(gpsig_ctx* c, const gpsig_params* p, bool apply_scaling, const void* X, int64_t N, int L, const double* w,
                 int squared, int id_dlev, int id_fac, const void** fac) {
    const int d_eff = p->num_features * ((apply_scaling ? p->num_lags : 0) + 1);
    SeqPlanned pl;
    const int rc = plan_seq(c, p, d_eff, L, &pl);
    if (rc == GPSIG_ERR_UNSUPPORTED && generic_ok(p)) {
        void* dl;
        CHK(ensure(c, id_dlev, sizeof(double) * size_t(N) * (p->num_levels + 1) + 8, &dl));
        CHK(generic_levels(c, p, apply_scaling, X, X, N, N, L, L, true, static_cast<double*>(dl), 1, p->num_levels + 1, 0));
        return make_factors(c, dl, N, p->num_levels + 1, w, p->jitter, id_fac, fac, squared);
    }
    CHK(rc);
    const void* rec;
    SeqGeom g;
    CHK(make_records(c, p, apply_scaling, pl, X, N, L, B_REC0, &rec, &g));
    const void* dl;
    CHK(diag_levels(c, p, pl, rec, g, N, id_dlev, &dl));
    return make_factors(c, dl, N, p->num_levels + 1, w, p->jitter, id_fac, fac, squared);
}

// diagonal levels of N sequences into out (M+1, N): the wavefront kernel where it is built for the shape, else the fallback
static int diag_levels_mn(gpsig_ctx* c, const gpsig_params* p, bool apply_scaling, const void* X, int64_t N, int L, void* out) {
    const int d_eff = p->num_features * ((apply_scaling ? p->num_lags : 0) + 1);
    SeqPlanned pl;
    const int rc = plan_seq(c, p, d_eff, L, &pl);
    // (option wide = 1: the wide route wherever built -- generic_levels takes it)
    if ((rc == GPSIG_ERR_UNSUPPORTED || (rc == GPSIG_OK && c->wide == 1 && wide_lat_available(c, p, L, L))) && generic_ok(p))
        return generic_levels(c, p, apply_scaling, X, X, N, N, L, L, true, static_cast<double*>(out), N, 1, 0);
    CHK(rc);
    const void* rec;
    SeqGeom g;
    CHK(make_records(c, p, apply_scaling, pl, X, N, L, B_REC0, &rec, &g));
    SeqRun r;
    memset(&r, 0, sizeof(r));
    r.xrec = rec; r.yrec = rec; r.gx = g; r.gy = g; r.N1 = N; r.N2 = N;
    r.out = out; r.si = 1; r.sj = 0; r.sm = N; r.pred = PRED_DIAG; r.timed = true;
    return launch_seq(c, p, pl, r);
}

// Core of K / _K_seq on device pointers.  raw: no scaling, no normalisation, no weights (levels out).
// x_squared: X-side factor 1/(diag+jitter) instead of 1/sqrt(diag+jitter) (K_seq_n_seq_covs quirk, kernels.py:713+:750).
static int seq_K_device(gpsig_ctx* c, const gpsig_params* p, bool raw, const void* X, const void* X2, int64_t N1, int64_t N2,
                 int L1, int L2, int return_levels, void* out, bool timed, int x_squared = 0, int64_t row_begin = 0,
                 int64_t row_end = 0, int compact = 0) {
    if (L1 < 1 || L2 < 1) return fail(c, GPSIG_ERR_INVALID, "sequence length must be >= 1");
    {       // the linear / cosine kernel's Gram as one contraction of explicit level features, where that is the cheaper evaluation (float32 calls: computed in float64)
        bool done = false;
        CHK(sig_features_K(c, p, raw, X, X2, N1, N2, L1, L2, return_levels, out, timed, x_squared, row_begin, row_end, compact, &done));
        if (done) return GPSIG_OK;
    }
    const bool sym = X2 == nullptr;
    const int M1 = p->num_levels + 1;
    const int d_eff = p->num_features * ((raw ? 0 : p->num_lags) + 1);
    // register-resident ("y") side: X2 by default; the shorter one if the lengths differ
    bool swap = false;
    if (!sym && L1 < L2) swap = true;
    SeqPlanned pl;
    const int64_t pairs_hint = sym ? N1 * (N1 / 2 + 1) : N1 * N2;
    int rc = plan_seq(c, p, d_eff, sym ? L1 : (swap ? L1 : L2), &pl, pairs_hint);
    if (rc != GPSIG_OK && !sym) {       // maybe the other side fits
        swap = !swap;
        rc = plan_seq(c, p, d_eff, swap ? L1 : L2, &pl, pairs_hint);
    }
    if (rc == GPSIG_OK && c->wide == 1 && generic_ok(p) && row_end == 0 && wide_lat_available(c, p, L1, L2)) rc = GPSIG_ERR_UNSUPPORTED;   // (option wide = 1)
    // 17 .. 32 columns, first order: the exact-shape kernels' 32-column instances hold four lattice columns x 32 features per lane at one wavefront per
    // SIMD -- a Gram of 384 sequences of 50 x 17 takes 10.8 ms there, 4.5 through the wide route's dgemm + lattice sweeps (tools/probe_shapes_rbf.py)
    if (rc == GPSIG_OK && c->wide < 0 && sizeof(TT) == 8 && d_eff > 16 && pairs_hint >= 4096 && generic_ok(p) && row_end == 0 && !(p->order > 1 && p->num_levels > 1) &&
        wide_lat_available(c, p, L1, L2))
        rc = GPSIG_ERR_UNSUPPORTED;
    if (rc == GPSIG_ERR_UNSUPPORTED && generic_ok(p) && row_end == 0) {     // any-shape fallback (wide route where built; else orders of magnitude slower per pair)
        if (timed) { c->t_launches += 0; }
        return seq_K_generic(c, p, raw, X, X2, N1, N2, L1, L2, return_levels, out, x_squared);
    }
    if (rc != GPSIG_OK) return rc;
    const void *fa = nullptr, *fb = nullptr;
    double jitter_diag = 0.0;
    if (!raw) {
        const double* w;
        CHK(upload_weights(c, p, &w));
        if (p->normalization) {
            CHK(side_factors(c, p, true, X, N1, L1, w, x_squared, B_DLEV0, B_FAC0, &fa));
            if (sym) CHK(make_factors(c, c->buf[B_DLEV0].p, N1, M1, nullptr, p->jitter, B_FAC1, &fb));
            else CHK(side_factors(c, p, true, X2, N2, L2, nullptr, 0, B_DLEV1, B_FAC1, &fb));
            if (sym) jitter_diag = p->jitter;        // kernels.py:431 (symmetric) vs :463-464 (cross: diagonals only)
        } else {
            CHK(make_factors(c, nullptr, N1, M1, w, 0.0, B_FAC0, &fa));
        }
    }
    const void *rec1, *rec2;
    SeqGeom g1, g2;
    CHK(make_records(c, p, !raw, pl, X, N1, L1, B_REC0, &rec1, &g1));
    if (sym) { rec2 = rec1; g2 = g1; }
    else CHK(make_records(c, p, !raw, pl, X2, N2, L2, B_REC1, &rec2, &g2));
    SeqRun r;
    memset(&r, 0, sizeof(r));
    const int64_t Ncols = sym ? N1 : N2;
    r.out = out; r.sm = N1 * Ncols;
    r.sum_levels = return_levels ? 0 : 1;
    if (raw) r.sum_levels = 0;
    r.jitter_diag = jitter_diag;
    r.pred = sym ? PRED_CIRCULANT : PRED_ALL;
    r.mirror = sym ? 1 : 0;
    r.timed = timed;
    if (row_end > 0) {   // owned-row block of the symmetric Gram: row = y index, no mirror, block-local row offset
        r.xrec = rec1; r.yrec = rec1; r.gx = g1; r.gy = g1; r.N1 = N1;
        r.N2 = row_end;          // y indices >= row_end belong to the next block: never loaded, never emitted
        r.si = 1; r.sj = compact ? N1 / 2 + 1 : N1; r.ax = fa; r.by = fb; r.mirror = 0; r.compact = compact;
        r.y_begin = row_begin; r.y_end = row_end;
        r.out = static_cast<TT*>(out) - row_begin * r.sj;
        return launch_seq(c, p, pl, r);
    }
    if (!swap) {   // x = X (rows of the output), y = X2 (columns)
        r.xrec = rec1; r.yrec = rec2; r.gx = g1; r.gy = g2; r.N1 = N1; r.N2 = Ncols;
        r.si = Ncols; r.sj = 1; r.ax = fa; r.by = fb;
    } else {       // x = X2 (columns), y = X (rows)
        r.xrec = rec2; r.yrec = rec1; r.gx = g2; r.gy = g1; r.N1 = N2; r.N2 = N1;
        r.si = 1; r.sj = N2; r.ax = fb; r.by = fa;
    }
    return launch_seq(c, p, pl, r);
}

static size_t seq_out_elems(const gpsig_params* p, int64_t N1, int64_t N2, int levels) {
    return size_t(N1) * size_t(N2) * (levels ? size_t(p->num_levels + 1) : 1);
}

// ---- tensors ---------------------------------------------------------------------------------------
static int prep_tensors(gpsig_ctx* c, const gpsig_params* p, bool apply_scaling, const void* Zdev, int64_t Tn, int E,
                 const void** ZT, const void** ZS) {
    const int lt = p->num_levels * (p->num_levels + 1) / 2;
    ScaleParams s;
    CHK(scale_params(c, p, apply_scaling, &s));
    const int d_eff = s.d_eff();
    void *zt, *zs;
    CHK(ensure(c, B_ZT, sizeof(TT) * size_t(Tn) * d_eff * lt * E + 8, &zt));
    CHK(ensure(c, B_ZS, sizeof(TT) * size_t(Tn) * lt * E + 8, &zs));
    if (Tn > 0) {
        hipLaunchKernelGGL(prep_tensors_kernel<TT>, dim3(grid_for(Tn * lt * E)), dim3(256), 0, c->stream,
                           static_cast<const TT*>(Zdev), lt, Tn, E, s, static_cast<TT*>(zt), static_cast<TT*>(zs));
        HIPCHK(c, hipGetLastError());
    }
    *ZT = zt; *ZS = zs;
    return GPSIG_OK;
}

static int tens_gram_device(gpsig_ctx* c, const gpsig_params* p, bool raw, const void* Z, int64_t Tn, int increments,
                     int return_levels, void* out) {
    const int E = increments ? 2 : 1;
    if (sizeof(TT) == 8 && Tn > 0) {        // wide state spaces (wide_api.hip): beyond 12 columns, or wherever built when the option says so
        ScaleParams s;
        CHK(scale_params(c, p, !raw, &s));
        if (wide_tens_available(c, p, Tn) && (c->wide == 1 || s.d_eff() > 12)) {
            const double* w = nullptr;
            if (!raw) CHK(upload_weights(c, p, &w));
            return wide_tens_forward(c, p, s, s.d_eff(), static_cast<const double*>(Z), Tn, increments, w, (raw || return_levels) ? 0 : 1, static_cast<double*>(out));
        }
    }
    const void *ZT, *ZS;
    CHK(prep_tensors(c, p, !raw, Z, Tn, E, &ZT, &ZS));
    TensGramArgs A;
    memset(&A, 0, sizeof(A));
    A.ZT = ZT; A.ZS = ZS; A.Tn = Tn; A.M = p->num_levels; A.d_eff = p->num_features * ((raw ? 0 : p->num_lags) + 1);
    A.E = E; A.kind = p->base_kernel;
    base_p(p, &A.p0, &A.p1);
    CHK(spectral_table(c, p, &A.spec));
    A.w = nullptr;
    if (!raw) CHK(upload_weights(c, p, &A.w));
    A.out = out;
    A.sum_levels = (raw || return_levels) ? 0 : 1;
    if (Tn > 0) {
        const int lt = A.M * (A.M + 1) / 2;
        const int RSZ = tens_gram_tile_stride(A.d_eff, lt, E);
        const size_t lds = sizeof(TT) * 2 * TENS_TILE * size_t(RSZ);
        if (p->base_kernel != GPSIG_BASE_SPECTRAL && lds <= 64 * 1024 && c->tens_tile != 0) {       // 16 x 16 tiles, tensors staged in LDS
            const unsigned nb = unsigned((Tn + TENS_TILE - 1) / TENS_TILE);
            hipLaunchKernelGGL(tens_gram_tile_kernel<TT>, dim3(nb, nb), dim3(TENS_TILE * TENS_TILE), lds, c->stream, A, RSZ);
        } else {
            hipLaunchKernelGGL(tens_gram_kernel<TT>, dim3(grid_for(Tn * Tn)), dim3(256), 0, c->stream, A);
        }
        HIPCHK(c, hipGetLastError());
    }
    return GPSIG_OK;
}

// fx: per-sequence factors (N, M+1) or NULL.
static int tens_vs_seq_lanet_device(gpsig_ctx* c, const gpsig_params* p, bool raw, const void* Zdev, const void* X, int64_t Tn,
                             int64_t N, int L, int increments, const void* fx, const double* w, int return_levels, void* out,
                             const TvsLaneTLaunchFn* fns, int ngroups) {
    const int M = p->num_levels, lt = M * (M + 1) / 2, E = increments ? 2 : 1;
    ScaleParams s;
    CHK(scale_params(c, p, !raw, &s));
    const int d_eff = s.d_eff();
    const int64_t Tpad = (Tn + 63) / 64 * 64;
    void *zl, *zn, *xs;
    CHK(ensure(c, B_ZL, sizeof(TT) * size_t(lt) * E * d_eff * Tpad + 8, &zl));
    CHK(ensure(c, B_ZN, sizeof(TT) * size_t(lt) * E * Tpad + 8, &zn));
    CHK(ensure(c, B_XT, sizeof(TT) * size_t(N) * L * d_eff + 8, &xs));
    hipLaunchKernelGGL(prep_tensors_lanet_kernel<TT>, dim3(grid_for(Tpad * lt * E)), dim3(256), 0, c->stream,
                       static_cast<const TT*>(Zdev), lt, Tn, Tpad, E, s, static_cast<TT*>(zl), static_cast<TT*>(zn));
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(prep_seq_scaled_kernel<TT>, dim3(grid_for(N * int64_t(L) * d_eff)), dim3(256), 0, c->stream,
                       static_cast<const TT*>(X), N, L, s, static_cast<TT*>(xs));
    HIPCHK(c, hipGetLastError());
    TvsLaneTArgs A;
    memset(&A, 0, sizeof(A));
    A.XS = xs; A.ZL = zl; A.ZN = zn; A.N = N; A.Tn = Tn; A.Tpad = Tpad;
    A.L = L; A.d_eff = d_eff; A.kind = p->base_kernel; A.difference = p->difference; A.order = p->order; A.M = M;
    base_p(p, &A.p0, &A.p1);
    A.fx = fx; A.w = w; A.out = out; A.sum_levels = (raw || return_levels) ? 0 : 1;
    hipEvent_t e0, e1;
    bool timed;
    CHK(timing_begin(c, &e0, &e1, &timed));
    for (int g = 0; g < ngroups; ++g) HIPCHK(c, fns[g](A, c->stream));
    if (timed) {
        HIPCHK(c, hipEventRecord(e1, c->stream));
        c->t_launches += 1;
        c->t_pairs += Tn * N;
    }
    return GPSIG_OK;
}

// Kzx through the tile kernel (tvs_tile_kernel.hpp): float64, order 1, many tensors.  Returns GPSIG_OK with *done = false
// when the kernel is not built for the shape (the caller goes on to the older mappings).
static int tens_vs_seq_tile_device(gpsig_ctx* c, const gpsig_params* p, bool raw, const void* Zdev, const void* X, int64_t Tn,
                                   int64_t N, int L, int increments, const void* fx, const double* w, int return_levels, void* out,
                                   bool* done) {
    *done = false;
    if (sizeof(TT) != 8 || p->base_kernel == GPSIG_BASE_SPECTRAL || c->tvs_tile == 0) return GPSIG_OK;
    const int M = p->num_levels, lt = M * (M + 1) / 2;
    const bool ho = p->order > 1 && M > 1;                   // higher-order chains (signature_algs.py:129-160): the RBF instances of tvs_tile_inst_ho.hip
    if (ho && p->base_kernel != GPSIG_BASE_RBF) return GPSIG_OK;
    ScaleParams s;
    CHK(scale_params(c, p, !raw, &s));
    const int d_eff = s.d_eff();
    const int D = tvs_tile_width(d_eff);
    if (D == 0) return GPSIG_OK;
    const int kind = p->base_kernel == GPSIG_BASE_LINEAR ? BASE_LINEAR : (p->base_kernel == GPSIG_BASE_RBF ? BASE_RBF :
                     (tvs_is_matern(p->base_kernel) ? p->base_kernel : -1));              // (the enums of include/gpsig_hip.h and seq_core.hpp agree)
    const bool collapse = increments && kind == BASE_LINEAR;              // <x, z1> - <x, z0> = <x, z1 - z0>
    const int E = (increments && !collapse) ? 2 : 1;
    // level sets: the planner's count, or the option's (A/B runs, tests) -- not for the Matern families, which are built for the planner's count only
    const int NW = (c->tvs_tile_nw > 0 && !tvs_is_matern(kind)) ? c->tvs_tile_nw : tvs_tile_waves(M, D, E, kind);
    TvsTileLaunchFn fn = NW > 0 ? (ho ? tvs_tile_lookup_ho(M, NW, D, E == 2) : tvs_tile_lookup(M, NW, D, E == 2, kind)) : nullptr;
    if (!fn) return GPSIG_OK;
    const int RS = tvs_row_stride(D);
    const bool sum_levels = !(raw || return_levels);
    const size_t lds = tvs_tile_lds_bytes(M, NW, sum_levels, E == 2);
    if (lds > 64 * 1024) return GPSIG_OK;
    const int64_t Tpad = (Tn + 63) / 64 * 64, TB = Tpad / 64;
    if (N > (int64_t(1) << 30)) return GPSIG_OK;                          // (the item counters are 32-bit)
    const bool matern = tvs_is_matern(kind);
    const double pre = kind == BASE_RBF ? tvs_rbf_prescale(E == 2) : (matern ? tvs_matern_prescale(kind, E == 2) : 1.0);
    const double pre_z = matern ? -2.0 * pre : pre;                        // (tvs_tile_kernel.hpp: the Matern components carry a factor -2)
    const int rows_are_increments = kind == BASE_LINEAR && p->difference;
    void *zl, *zn, *xr;
    CHK(ensure(c, B_ZL, sizeof(double) * size_t(lt) * E * D * Tpad + 8, &zl));
    CHK(ensure(c, B_ZN, sizeof(double) * size_t(lt) * E * Tpad + 8, &zn));
    const int64_t nrows = N * int64_t(L) + 1;                             // (one row behind the last sequence: the sweep asks one row ahead)
    CHK(ensure(c, B_XT, sizeof(double) * size_t(nrows) * RS + 64, &xr));
    void* tq;
    CHK(ensure(c, B_TQ, sizeof(int32_t) * size_t(TB) + 8, &tq));
    hipLaunchKernelGGL(prep_tensors_tile_kernel, dim3(grid_for(Tpad * lt * E)), dim3(256), 0, c->stream,
                       static_cast<const double*>(Zdev), lt, Tn, Tpad, increments ? 2 : 1, collapse ? 1 : 0, pre_z, s, D,
                       static_cast<double*>(zl), static_cast<double*>(zn), static_cast<int32_t*>(tq));
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(prep_seq_tile_rows_kernel, dim3(grid_for(nrows * RS)), dim3(256), 0, c->stream,
                       static_cast<const double*>(X), N, L, s, pre, rows_are_increments, D, RS, static_cast<double*>(xr));
    HIPCHK(c, hipGetLastError());
    TvsTileArgs A;
    memset(&A, 0, sizeof(A));
    A.XR = xr; A.ZL = zl; A.ZN = zn; A.N = N; A.Tn = Tn; A.Tpad = Tpad;
    A.L = L; A.d = d_eff; A.kind = p->base_kernel; A.difference = p->difference; A.M = M;
    A.order = p->order;
    A.queue = static_cast<int32_t*>(tq);          // (the launch plans the items: it knows the instance's occupancy)
    base_p(p, &A.p0, &A.p1);
    A.fx = fx; A.w = w; A.out = out; A.sum_levels = sum_levels ? 1 : 0;
    A.aux = c->tvs_aux_out;
    if (A.aux) c->tvs_aux_written = true;
    hipEvent_t e0, e1;
    bool timed;
    CHK(timing_begin(c, &e0, &e1, &timed));
    HIPCHK(c, fn(A, lds, c->stream, c->num_cus));
    if (timed) {
        HIPCHK(c, hipEventRecord(e1, c->stream));
        c->t_launches += 1;
        c->t_pairs += Tn * N;
    }
    *done = true;
    return GPSIG_OK;
}

// Kzx on device pointers.  Zdev: the caller's (lt, T, E, d') tensor array; ZT/ZS: its sequence-lane preparation.
static bool tvs_features_plan(gpsig_ctx* c, const gpsig_params* p, int64_t Tn, int64_t N, int L) {
    const bool cosine = p->base_kernel == GPSIG_BASE_COSINE;
    const int M = p->num_levels;
    if (sizeof(TT) != 8 || c->tvs_features == 0 || c->capturing || !(p->base_kernel == GPSIG_BASE_LINEAR || cosine)) return false;
    if (M < 2 || M > 8 || p->order < 1 || p->order > M || Tn <= 0 || N <= 0 || Tn > 0x3fffffff || N > 0x3fffffff) return false;
    const int d = p->num_features * (p->num_lags + 1);
    SigFeatLaunchFn ffn = sig_feat_lookup(d, M);
    if (!ffn || (p->difference ? L - 1 : L) < 1) return false;
    if (sig_features_lds_bytes(d, M, L) > 150 * 1024) return false;
    for (int m = 0; m <= M; ++m)
        if (!(p->sigma * p->variances[m] >= 0.0)) return false;             // (the weights go in squared: sqrt(w^2 / (diag + jitter)))
    const int64_t F = sig_feature_count(d, M), ld = (F + 1 + 15) / 16 * 16;
    if (sizeof(double) * size_t(ld) * (size_t(N) + size_t(Tn)) > (size_t(16) << 30)) return false;
    if (c->tvs_features < 0) {
        // rates as measured at BASELINE configs[2] (512 x 16,384, L = 50, d = 6, M = 4): the tile kernel 1.05 ms, here 0.18 + 0.42 ms
        const int lt = M * (M + 1) / 2;
        const double t_tile = 40e-6 + double(Tn) * double(N) * L * lt / 4.0e12;
        const double t_seq = 60e-6 * double(sig_ipow(d, M)) / 32768.0, rounds = ceil(double(N) / 256.0);
        const double t_feat = 80e-6 + (rounds * t_seq > 15e-6 ? rounds * t_seq : 15e-6) + 2.0 * double(Tn) * double(N) * double(ld) / 50e12;
        if (!(t_feat < t_tile)) return false;
    }
    return true;
}
// Kzx = sum_m w_m / sqrt(K_m(x, x) + jitter) K_m(z, x) (kernels.py:572-588) of the linear / cosine kernel as ONE product of the tensors' and the
// sequences' level features (rocBLAS dgemm): the weights and the normalisation ride on the sequences' features.  *done = false: the tile kernel.
static int tens_vs_seq_features_device(gpsig_ctx* c, const gpsig_params* p, const void* ZT, const void* ZS, const void* X, int64_t Tn, int64_t N,
                                       int L, int increments, const double* w, void* out, bool* done) {
    *done = false;
    if (!tvs_features_plan(c, p, Tn, N, L)) return GPSIG_OK;
    const bool cosine = p->base_kernel == GPSIG_BASE_COSINE;
    const int M = p->num_levels, d = p->num_features * (p->num_lags + 1);
    SigFeatLaunchFn ffn = sig_feat_lookup(d, M);
    const int64_t F = sig_feature_count(d, M), ld = (F + 1 + 15) / 16 * 16;
    void *phi, *zf, *w2;
    CHK(ensure(c, B_SF0, sizeof(double) * size_t(ld) * N + 64, &phi));
    CHK(ensure(c, B_SF1, sizeof(double) * size_t(ld) * Tn + 64, &zf));
    CHK(ensure(c, B_TW2, sizeof(double) * 16, &w2));
    c->sf_valid = false;                          // (B_SF0 no longer holds what "sig_features_keep" remembers)
    hipLaunchKernelGGL(square_small_kernel, dim3(1), dim3(64), 0, c->stream, w, M + 1, static_cast<double*>(w2));
    HIPCHK(c, hipGetLastError());
    ScaleParams s;
    CHK(scale_params(c, p, true, &s));
    SigFeatArgs A;
    memset(&A, 0, sizeof(A));
    A.X = static_cast<const double*>(X); A.N = N; A.L = L; A.difference = p->difference ? 1 : 0; A.P = s;
    A.w = static_cast<const double*>(w2); A.normalize = p->normalization ? 1 : 0; A.jitter = p->jitter; A.Phi = static_cast<double*>(phi); A.ld = ld;
    A.dlev = nullptr; A.order = p->order; A.unit_points = cosine ? 1 : 0; A.norm_squared = 0; A.natural_order = 1;
    {
        // (workgroups of one or two wavefronts -- small d^M -- are latency-bound at 16 per CU: up to 16,384 of them; 0.667 -> 0.647 ms at configs[2])
        const int64_t cap = sig_threads(d, M) <= 128 ? 16384 : 4096;
        hipError_t e = ffn(A, unsigned(N < cap ? N : cap), sig_features_lds_bytes(d, M, L), c->stream);
        if (e != hipSuccess) return fail(c, GPSIG_ERR_HIP, "sig_features_kernel: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(tens_level_features_kernel, dim3(grid_for(Tn * ld)), dim3(256), 0, c->stream, static_cast<const double*>(ZT),
                       static_cast<const double*>(ZS), M * (M + 1) / 2, increments ? 2 : 1, d, Tn, M, cosine ? 1 : 0, ld, static_cast<double*>(zf));
    HIPCHK(c, hipGetLastError());
    hipEvent_t e0, e1;
    bool timed;
    CHK(timing_begin(c, &e0, &e1, &timed));
    // row-major out (T, N) = Zf (T, ld) Phi (N, ld)^T  ==  column-major out^T (N x T) = Phi_cm^T (N x ld) Zf_cm (ld x T)
    std::string err;
    if (!solver_dgemm(&c->blas_handle, c->stream, true, false, int(N), int(Tn), int(ld), 1.0, static_cast<const double*>(phi), int(ld),
                      static_cast<const double*>(zf), int(ld), 0.0, static_cast<double*>(out), int(N), &err))
        return fail(c, GPSIG_ERR_HIP, "%s", err.c_str());
    if (timed) {
        HIPCHK(c, hipEventRecord(e1, c->stream));
        c->t_launches += 1;
        c->t_pairs += Tn * N;
        c->t_kernel = "tvs_features_dgemm";
        c->t_flops += 2.0 * double(Tn) * double(N) * double(ld);
    }
    *done = true;
    return GPSIG_OK;
}

static int tens_vs_seq_device(gpsig_ctx* c, const gpsig_params* p, bool raw, const void* Zdev, const void* ZT, const void* ZS,
                       const void* X, int64_t Tn, int64_t N, int L, int increments, const void* fx, const double* w,
                       int return_levels, void* out) {
    const int M = p->num_levels;
    if (!raw && !return_levels && w) {      // the linear / cosine kernel's weighted level sum: one product of level features
        bool done = false;
        CHK(tens_vs_seq_features_device(c, p, ZT, ZS, X, Tn, N, L, increments, w, out, &done));
        if (done) return GPSIG_OK;
    }
    // wide state spaces (wide_api.hip): beyond the tile kernel's 8 columns, or wherever built when the option says so
    auto wide = [&](bool* done) -> int {
        *done = false;
        if (sizeof(TT) != 8) return GPSIG_OK;
        ScaleParams s;
        CHK(scale_params(c, p, !raw, &s));
        const int d_eff = s.d_eff();
        // beyond the tile kernel's 8 columns -- and, at 5 .. 8 columns with increments, launches of few sequences: the tile kernel's two level sets sweep a
        // 16-sequence tile one after the other, 8 us per time step whatever the count, where the chains here take 2.4 (ECG's shape: 1.23 -> 0.36 ms,
        // profiles/r06_ab_small_widths.txt); the reverse tile kernel continues from the same chain totals (at 7 / 8 columns the forward pass is what the
        // older mappings take -- 0.83 against 0.68 ms at UWave's shape --, the reverse pass saves its own forward sweep: 2.05 -> 1.59; option
        // wide_few_cols: the widest state space of this rule, A/B runs)
        const bool few = increments && d_eff > 4 && d_eff <= (c->wide_few_cols > 0 ? c->wide_few_cols : 8) && N <= 256 && p->base_kernel == GPSIG_BASE_RBF;
        if (!wide_tvs_available(c, p, d_eff, Tn, N, L) || !(c->wide == 1 || d_eff > 8 || (few && c->wide != 0 && c->tvs_tile != 1))) return GPSIG_OK;
        void* xs;
        CHK(ensure(c, B_XT, sizeof(double) * size_t(N) * L * d_eff + 8, &xs));
        hipLaunchKernelGGL(prep_seq_scaled_kernel<double>, dim3(grid_for(N * int64_t(L) * d_eff)), dim3(256), 0, c->stream,
                           static_cast<const double*>(X), N, L, s, static_cast<double*>(xs));
        HIPCHK(c, hipGetLastError());
        double* aux = c->tvs_aux_out;
        if (aux) c->tvs_aux_written = true;
        CHK(wide_tvs_forward(c, p, s, d_eff, static_cast<const double*>(Zdev), static_cast<const double*>(xs), Tn, N, L, increments,
                             static_cast<const double*>(fx), w, (raw || return_levels) ? 0 : 1, static_cast<double*>(out), aux));
        *done = true;
        return GPSIG_OK;
    };
    {
        bool done = false;        // (the lambda declines where the tile kernel is the better choice)
        CHK(wide(&done));
        if (done) return GPSIG_OK;
    }
    if (N > 0 && Tn > 0 && c->tens_lanes != 0 && (Tn >= 32 || c->tens_lanes == 1 || c->tvs_tile == 1)) {
        bool done = false;
        CHK(tens_vs_seq_tile_device(c, p, raw, Zdev, X, Tn, N, L, increments, fx, w, return_levels, out, &done));
        if (done) return GPSIG_OK;
    }
    if (N > 0 && Tn > 0 && N <= 65535 && c->tens_lanes != 0 && (Tn >= 32 || c->tens_lanes == 1) && p->base_kernel != GPSIG_BASE_SPECTRAL &&
        size_t(L) * scale_of(p, !raw).d_eff() * sizeof(TT) <= 48 * 1024) {
        TvsLaneTLaunchFn fns[8];
        int ng = 0;
        if (sizeof(TT) == 8 && tvs_lanet_plan(M, scale_of(p, !raw).d_eff(), increments != 0, fns, &ng))
            return tens_vs_seq_lanet_device(c, p, raw, Zdev, X, Tn, N, L, increments, fx, w, return_levels, out, fns, ng);
    }
    const int64_t Npad = (N + 63) / 64 * 64;
    ScaleParams sx;
    CHK(scale_params(c, p, !raw, &sx));
    const int d_eff = sx.d_eff();
    void* xt;
    CHK(ensure(c, B_XT, sizeof(TT) * size_t(L) * d_eff * Npad + 8, &xt));
    if (N > 0) {
        hipLaunchKernelGGL(prep_seq_timemajor_kernel<TT>, dim3(grid_for(int64_t(L) * d_eff * Npad)), dim3(256), 0,
                           c->stream, static_cast<const TT*>(X), N, Npad, L, sx, static_cast<TT*>(xt));
        HIPCHK(c, hipGetLastError());
    }
    const int lt = M * (M + 1) / 2;
    const int E = increments ? 2 : 1;
    int NTT = (lt * (2 + E) * 2 <= 100) ? 2 : 1;
    TvsLaunchFn fn = tvs_lookup(M, NTT, increments != 0, sizeof(TT) == 4);
    if (!fn) return fail(c, GPSIG_ERR_UNSUPPORTED, "tensor-vs-sequence kernel is built for num_levels <= 8 (got %d)", M);
    TvsArgs A;
    memset(&A, 0, sizeof(A));
    A.XT = xt; A.ZT = ZT; A.ZS = ZS; A.N = N; A.Npad = Npad; A.Tn = Tn;
    A.L = L; A.d_eff = d_eff; A.kind = p->base_kernel; A.difference = p->difference; A.order = p->order;
    base_p(p, &A.p0, &A.p1);
    CHK(spectral_table(c, p, &A.spec));
    A.fx = fx; A.w = w; A.out = out; A.sum_levels = (raw || return_levels) ? 0 : 1;
    if (N > 0 && Tn > 0) {
        hipEvent_t e0, e1;
        bool timed;
        CHK(timing_begin(c, &e0, &e1, &timed));
        HIPCHK(c, fn(A, c->stream));
        if (timed) {
            HIPCHK(c, hipEventRecord(e1, c->stream));
            c->t_launches += 1;
            c->t_pairs += Tn * N;
        }
    }
    return GPSIG_OK;
}

// per-sequence 1/sqrt(diag + jitter) factors (N, M+1) of sequences X, into B_FAC1
static int seq_diag_factors(gpsig_ctx* c, const gpsig_params* p, const void* X, int64_t N, int L, const void** fac) {
    return side_factors(c, p, true, X, N, L, nullptr, 0, B_DLEV0, B_FAC1, fac);
}

static int e_seq_gram_levels(gpsig_ctx* c, const gpsig_params* p, const void* X, const void* X2, int64_t N1, int64_t N2,
                          int32_t L1, int32_t L2, void* out) {
    ENTER(c, p);
    const int d = p->num_features * (p->num_lags + 1);
    gpsig_params q = *p;
    q.num_features = d; q.num_lags = 0;      // raw entry point: columns are taken as they come
    if (d > MAX_FEATURES && !(sizeof(TT) == 8 && d <= MAX_FEATURES_WIDE && wide_lat_available(c, p, L1, X2 ? L2 : L1)))
        return fail(c, GPSIG_ERR_UNSUPPORTED, "d=%d too large", d);
    const void *dX, *dX2 = nullptr;
    CHK(in_dev(c, B_IN0, X, sizeof(TT) * size_t(N1) * L1 * d, &dX));
    if (X2) CHK(in_dev(c, B_IN1, X2, sizeof(TT) * size_t(N2) * L2 * d, &dX2));
    const int64_t Nc = X2 ? N2 : N1;
    const size_t ob = sizeof(TT) * seq_out_elems(p, N1, Nc, 1);
    void* dout;
    CHK(out_dev(c, B_OUT0, out, ob, &dout));
    CHK(seq_K_device(c, &q, true, dX, dX2, N1, Nc, L1, X2 ? L2 : L1, 1, dout, true));
    CHK(out_done(c, out, dout, ob));
    return finish(c);
}

static int e_seq_diag_levels(gpsig_ctx* c, const gpsig_params* p, const void* X, int64_t N, int32_t L, void* out) {
    ENTER(c, p);
    const int d = p->num_features * (p->num_lags + 1);
    gpsig_params q = *p;
    q.num_features = d; q.num_lags = 0;
    if (d > MAX_FEATURES && !(sizeof(TT) == 8 && d <= MAX_FEATURES_WIDE && wide_lat_available(c, p, L, L)))
        return fail(c, GPSIG_ERR_UNSUPPORTED, "d=%d too large", d);
    const int M1 = p->num_levels + 1;
    const void* dX;
    CHK(in_dev(c, B_IN0, X, sizeof(TT) * size_t(N) * L * d, &dX));
    void* dout;
    CHK(out_dev(c, B_OUT0, out, sizeof(TT) * size_t(N) * M1, &dout));
    CHK(diag_levels_mn(c, &q, false, dX, N, L, dout));
    CHK(out_done(c, out, dout, sizeof(TT) * size_t(N) * M1));
    return finish(c);
}

static int e_tens_gram_levels(gpsig_ctx* c, const gpsig_params* p, const void* Z, int64_t T, int32_t increments, void* out) {
    ENTER(c, p);
    const int d = p->num_features * (p->num_lags + 1), lt = p->num_levels * (p->num_levels + 1) / 2, E = increments ? 2 : 1;
    if (d > MAX_FEATURES && !(sizeof(TT) == 8 && d <= MAX_FEATURES_WIDE && wide_tens_available(c, p, T)))
        return fail(c, GPSIG_ERR_UNSUPPORTED, "d=%d too large", d);
    const void* dZ;
    CHK(in_dev(c, B_IN0, Z, sizeof(TT) * size_t(lt) * T * E * d, &dZ));
    const size_t ob = sizeof(TT) * size_t(T) * T * (p->num_levels + 1);
    void* dout;
    CHK(out_dev(c, B_OUT0, out, ob, &dout));
    gpsig_params q = *p;
    q.num_features = d; q.num_lags = 0;      // raw entry point: columns are taken as they come
    CHK(tens_gram_device(c, &q, true, dZ, T, increments, 1, dout));
    CHK(out_done(c, out, dout, ob));
    return finish(c);
}

static int e_tens_vs_seq_levels(gpsig_ctx* c, const gpsig_params* p, const void* Z, const void* X, int64_t T, int64_t N,
                             int32_t L, int32_t increments, void* out) {
    ENTER(c, p);
    const int d = p->num_features * (p->num_lags + 1), lt = p->num_levels * (p->num_levels + 1) / 2, E = increments ? 2 : 1;
    // (beyond MAX_FEATURES columns only the wide route -- wide_api.hip -- takes the call)
    if (d > MAX_FEATURES && !(sizeof(TT) == 8 && d <= MAX_FEATURES_WIDE && wide_tvs_available(c, p, d, T, N, L)))
        return fail(c, GPSIG_ERR_UNSUPPORTED, "d=%d too large", d);
    const void *dZ, *dX;
    CHK(in_dev(c, B_IN0, Z, sizeof(TT) * size_t(lt) * T * E * d, &dZ));
    CHK(in_dev(c, B_IN1, X, sizeof(TT) * size_t(N) * L * d, &dX));
    const size_t ob = sizeof(TT) * size_t(T) * N * (p->num_levels + 1);
    void* dout;
    CHK(out_dev(c, B_OUT0, out, ob, &dout));
    gpsig_params q = *p;
    q.num_features = d; q.num_lags = 0;      // raw entry point: columns are taken as they come
    const void *ZT, *ZS;
    CHK(prep_tensors(c, &q, false, dZ, T, E, &ZT, &ZS));
    CHK(tens_vs_seq_device(c, &q, true, dZ, ZT, ZS, dX, T, N, L, increments, nullptr, nullptr, 1, dout));
    CHK(out_done(c, out, dout, ob));
    return finish(c);
}

// sum_m fac[n][m] level_m[t][n] of the raw levels (inputs as they come), the level sum taken inside the kernels' epilogue
static int e_tens_vs_seq_weighted(gpsig_ctx* c, const gpsig_params* p, const void* Z, const void* X, int64_t T, int64_t N, int32_t L,
                               int32_t increments, const void* fac, void* out, void* aux, int32_t* aux_written) {
    ENTER(c, p);
    const int d = p->num_features * (p->num_lags + 1), lt = p->num_levels * (p->num_levels + 1) / 2, E = increments ? 2 : 1;
    if (d > MAX_FEATURES && !(sizeof(TT) == 8 && d <= MAX_FEATURES_WIDE && wide_tvs_available(c, p, d, T, N, L)))
        return fail(c, GPSIG_ERR_UNSUPPORTED, "d=%d too large", d);
    if (!fac && N > 0) return fail(c, GPSIG_ERR_INVALID, "null factor array");
    const void *dZ, *dX, *dF;
    CHK(in_dev(c, B_IN0, Z, sizeof(TT) * size_t(lt) * T * E * d, &dZ));
    CHK(in_dev(c, B_IN1, X, sizeof(TT) * size_t(N) * L * d, &dX));
    CHK(in_dev(c, B_IN2, fac, sizeof(TT) * size_t(N) * (p->num_levels + 1), &dF));
    const size_t ob = sizeof(TT) * size_t(T) * N;
    void* dout;
    CHK(out_dev(c, B_OUT0, out, ob, &dout));
    gpsig_params q = *p;
    q.num_features = d; q.num_lags = 0; q.lengthscales = nullptr;      // raw entry point: columns are taken as they come
    const void *ZT, *ZS;
    CHK(prep_tensors(c, &q, false, dZ, T, E, &ZT, &ZS));
    if (aux_written) *aux_written = 0;
    // the chain totals for the reverse pass: float64, device pointers (they stay on the device between the two passes)
    c->tvs_aux_written = false;
    c->tvs_aux_out = (aux && sizeof(TT) == 8 && c->ptr_mode == GPSIG_PTR_DEVICE) ? static_cast<double*>(aux) : nullptr;
    const int rc = tens_vs_seq_device(c, &q, false, dZ, ZT, ZS, dX, T, N, L, increments, dF, nullptr, 0, dout);
    c->tvs_aux_out = nullptr;
    CHK(rc);
    if (aux_written) *aux_written = c->tvs_aux_written ? 1 : 0;
    CHK(out_done(c, out, dout, ob));
    return finish(c);
}

static int e_kernel_K(gpsig_ctx* c, const gpsig_params* p, const void* X, const void* X2, int64_t N1, int64_t N2, int32_t L1,
                   int32_t L2, int32_t return_levels, void* out) {
    ENTER(c, p);
    const int d = p->num_features;
    const void *dX, *dX2 = nullptr;
    CHK(in_dev(c, B_IN0, X, sizeof(TT) * size_t(N1) * L1 * d, &dX));
    if (X2) CHK(in_dev(c, B_IN1, X2, sizeof(TT) * size_t(N2) * L2 * d, &dX2));
    const int64_t Nc = X2 ? N2 : N1;
    const size_t ob = sizeof(TT) * seq_out_elems(p, N1, Nc, return_levels);
    void* dout;
    CHK(out_dev(c, B_OUT0, out, ob, &dout));
    CHK(seq_K_device(c, p, false, dX, dX2, N1, Nc, L1, X2 ? L2 : L1, return_levels, dout, true));
    CHK(out_done(c, out, dout, ob));
    return finish(c);
}

static int e_kernel_K_symm_rows(gpsig_ctx* c, const gpsig_params* p, const void* X, int64_t N, int32_t L, int64_t row_begin,
                             int64_t row_end, void* out_rows, int compact) {
    ENTER(c, p);
    if (row_begin < 0 || row_end > N || row_begin > row_end || ((row_begin % 4) != 0 && row_begin != row_end))
        return fail(c, GPSIG_ERR_INVALID, "bad row range [%lld, %lld) of %lld (row_begin must be a multiple of 4)",
                    (long long)row_begin, (long long)row_end, (long long)N);
    const void* dX;
    CHK(in_dev(c, B_IN0, X, sizeof(TT) * size_t(N) * L * p->num_features, &dX));
    const size_t ob = sizeof(TT) * size_t(row_end - row_begin) * (compact ? N / 2 + 1 : N);
    void* dout;
    CHK(out_dev(c, B_OUT0, out_rows, ob, &dout));
    if (row_end > row_begin) {
        CHK(seq_K_device(c, p, false, dX, nullptr, N, N, L, L, 0, dout, true, 0, row_begin, row_end, compact));
    } else {
        // an empty block still reports whether the shape is one the row-block kernels take: every rank of a sharded evaluation
        // must reach the same verdict, also the one that owns no rows -- so it asks the same routes in the same order as seq_K_device
        // does for a block with rows: the feature contraction first (it takes shapes the pair kernels refuse), then the pair kernels
        bool done = false;
        CHK(sig_features_K(c, p, false, dX, nullptr, N, N, L, L, 0, dout, false, 0, row_begin, row_end, compact, &done, true));
        if (!done) {
            SeqPlanned pl;
            CHK(plan_seq(c, p, p->num_features * (p->num_lags + 1), L, &pl, N * (N / 2 + 1)));
        }
    }
    CHK(out_done(c, out_rows, dout, ob));
    return finish(c);
}

static int e_symmetrize_owned_rows(gpsig_ctx* c, int32_t dtype, const void* half, int64_t N, void* out) {
    if (!c) return GPSIG_ERR_INVALID;
    if (half == out) return fail(c, GPSIG_ERR_INVALID, "symmetrize needs distinct buffers");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t b = sizeof(TT) * size_t(N) * N;
    const void* dh;
    CHK(in_dev(c, B_IN0, half, b, &dh));
    void* dout;
    CHK(out_dev(c, B_OUT0, out, b, &dout));
    if (N > 0) {
        hipLaunchKernelGGL(symmetrize_owned_rows_kernel<TT>, dim3(grid_for(N * N)), dim3(256), 0, c->stream,
                           static_cast<const TT*>(dh), N, static_cast<TT*>(dout));
        HIPCHK(c, hipGetLastError());
    }
    CHK(out_done(c, out, dout, b));
    return finish(c);
}

static int e_symmetrize_compact_rows(gpsig_ctx* c, int32_t dtype, const void* half, int64_t N, void* out) {
    if (!c) return GPSIG_ERR_INVALID;
    if (half == out) return fail(c, GPSIG_ERR_INVALID, "symmetrize needs distinct buffers");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bi = sizeof(TT) * size_t(N) * (N / 2 + 1), bo = sizeof(TT) * size_t(N) * N;
    const void* dh;
    CHK(in_dev(c, B_IN0, half, bi, &dh));
    void* dout;
    CHK(out_dev(c, B_OUT0, out, bo, &dout));
    if (N > 0) {
        const int64_t tpr = (N + 63) / 64, ntiles = tpr * tpr;
        hipLaunchKernelGGL(symmetrize_compact_rows_kernel<TT>, dim3(unsigned(ntiles < 65536 ? ntiles : 65536)), dim3(64, 4), 0, c->stream,
                           static_cast<const TT*>(dh), N, static_cast<TT*>(dout));
        HIPCHK(c, hipGetLastError());
    }
    CHK(out_done(c, out, dout, bo));
    return finish(c);
}

static int e_kernel_Kdiag(gpsig_ctx* c, const gpsig_params* p, const void* X, int64_t N, int32_t L, int32_t return_levels, void* out) {
    ENTER(c, p);
    const int M1 = p->num_levels + 1;
    const size_t ob = sizeof(TT) * size_t(N) * (return_levels ? M1 : 1);
    void* dout;
    CHK(out_dev(c, B_OUT0, out, ob, &dout));
    const double* w;
    CHK(upload_weights(c, p, &w));
    void* tmp;
    CHK(ensure(c, B_TMP0, sizeof(TT) * size_t(N) * M1 + 8, &tmp));
    if (p->normalization) {
        // kernels.py:486-490: sigma * variances, no data touched
        if (N > 0) {
            hipLaunchKernelGGL(fill_kernel<TT>, dim3(grid_for(N * M1)), dim3(256), 0, c->stream, static_cast<TT*>(tmp), N * M1, TT(1));
            HIPCHK(c, hipGetLastError());
        }
    } else {
        const void* dX;
        CHK(in_dev(c, B_IN0, X, sizeof(TT) * size_t(N) * L * p->num_features, &dX));
        CHK(diag_levels_mn(c, p, true, dX, N, L, tmp));
    }
    if (N > 0) {
        hipLaunchKernelGGL(weight_levels_kernel<TT>, dim3(grid_for(N)), dim3(256), 0, c->stream,
                           static_cast<const TT*>(tmp), N, M1, w, return_levels ? 0 : 1, static_cast<TT*>(dout));
        HIPCHK(c, hipGetLastError());
    }
    CHK(out_done(c, out, dout, ob));
    return finish(c);
}

static int e_kernel_K_tens(gpsig_ctx* c, const gpsig_params* p, const void* Z, int64_t T, int32_t increments, int32_t return_levels, void* out) {
    ENTER(c, p);
    const int d = p->num_features * (p->num_lags + 1), lt = p->num_levels * (p->num_levels + 1) / 2, E = increments ? 2 : 1;
    const void* dZ;
    CHK(in_dev(c, B_IN0, Z, sizeof(TT) * size_t(lt) * T * E * d, &dZ));
    const size_t ob = sizeof(TT) * size_t(T) * T * (return_levels ? p->num_levels + 1 : 1);
    void* dout;
    CHK(out_dev(c, B_OUT0, out, ob, &dout));
    CHK(tens_gram_device(c, p, false, dZ, T, increments, return_levels, dout));
    CHK(out_done(c, out, dout, ob));
    return finish(c);
}

static int e_kernel_K_tens_vs_seq(gpsig_ctx* c, const gpsig_params* p, const void* Z, const void* X, int64_t T, int64_t N,
                               int32_t L, int32_t increments, int32_t return_levels, void* out) {
    ENTER(c, p);
    const int d = p->num_features * (p->num_lags + 1), lt = p->num_levels * (p->num_levels + 1) / 2, E = increments ? 2 : 1;
    const void *dZ, *dX;
    CHK(in_dev(c, B_IN0, Z, sizeof(TT) * size_t(lt) * T * E * d, &dZ));
    CHK(in_dev(c, B_IN1, X, sizeof(TT) * size_t(N) * L * p->num_features, &dX));
    const size_t ob = sizeof(TT) * size_t(T) * N * (return_levels ? p->num_levels + 1 : 1);
    void* dout;
    CHK(out_dev(c, B_OUT0, out, ob, &dout));
    const void* fx = nullptr;
    if (p->normalization && !(!return_levels && tvs_features_plan(c, p, T, N, L))) CHK(seq_diag_factors(c, p, dX, N, L, &fx));     // kernels.py:572-581
    const double* w;
    CHK(upload_weights(c, p, &w));
    const void *ZT, *ZS;
    CHK(prep_tensors(c, p, true, dZ, T, E, &ZT, &ZS));
    CHK(tens_vs_seq_device(c, p, false, dZ, ZT, ZS, dX, T, N, L, increments, fx, w, return_levels, dout));
    CHK(out_done(c, out, dout, ob));
    return finish(c);
}

static int e_kernel_K_tens_n_seq_covs(gpsig_ctx* c, const gpsig_params* p, const void* Z, const void* X, int64_t T, int64_t N,
                                   int32_t L, int32_t increments, int32_t full_X_cov, int32_t return_levels, void* Kzz,
                                   void* Kzx, void* Kxx) {
    ENTER(c, p);
    const int M1 = p->num_levels + 1;
    const int d = p->num_features * (p->num_lags + 1), lt = p->num_levels * (p->num_levels + 1) / 2, E = increments ? 2 : 1;
    const size_t lv = return_levels ? size_t(M1) : 1;
    const void *dZ, *dX;
    CHK(in_dev(c, B_IN0, Z, sizeof(TT) * size_t(lt) * T * E * d, &dZ));
    CHK(in_dev(c, B_IN1, X, sizeof(TT) * size_t(N) * L * p->num_features, &dX));
    const size_t bzz = sizeof(TT) * size_t(T) * T * lv, bzx = sizeof(TT) * size_t(T) * N * lv;
    const size_t bxx = sizeof(TT) * (full_X_cov ? size_t(N) * N : size_t(N)) * lv;
    void *dzz, *dzx, *dxx;
    CHK(out_dev(c, B_OUT0, Kzz, bzz, &dzz));
    CHK(out_dev(c, B_OUT1, Kzx, bzx, &dzx));
    CHK(out_dev(c, B_OUT2, Kxx, bxx, &dxx));
    const double* w;
    CHK(upload_weights(c, p, &w));
    // Kzz: never normalised (kernels.py:623, :641/:665)
    CHK(tens_gram_device(c, p, false, dZ, T, increments, return_levels, dzz));
    // Kzx: divided by sqrt(diag_x + jitter) when normalising (kernels.py:638 / :660) -- with full_X_cov the
    // diagonal of (Kxx + jitter*I) is the same number
    const void* fx = nullptr;
    if (p->normalization && !(!return_levels && tvs_features_plan(c, p, T, N, L))) CHK(seq_diag_factors(c, p, dX, N, L, &fx));
    const void *ZT, *ZS;
    CHK(prep_tensors(c, p, true, dZ, T, E, &ZT, &ZS));
    CHK(tens_vs_seq_device(c, p, false, dZ, ZT, ZS, dX, T, N, L, increments, fx, w, return_levels, dzx));
    if (full_X_cov) {
        CHK(seq_K_device(c, p, false, dX, nullptr, N, N, L, L, return_levels, dxx, true));   // kernels.py:630-640
    } else {
        void* tmp;
        CHK(ensure(c, B_TMP0, sizeof(TT) * size_t(N) * M1 + 8, &tmp));
        if (p->normalization) {   // kernels.py:661
            if (N > 0) {
                hipLaunchKernelGGL(fill_kernel<TT>, dim3(grid_for(N * M1)), dim3(256), 0, c->stream, static_cast<TT*>(tmp), N * M1, TT(1));
                HIPCHK(c, hipGetLastError());
            }
        } else {                  // kernels.py:653, :663
            CHK(diag_levels_mn(c, p, true, dX, N, L, tmp));
        }
        if (N > 0) {
            hipLaunchKernelGGL(weight_levels_kernel<TT>, dim3(grid_for(N)), dim3(256), 0, c->stream,
                               static_cast<const TT*>(tmp), N, M1, w, return_levels ? 0 : 1, static_cast<TT*>(dxx));
            HIPCHK(c, hipGetLastError());
        }
    }
    CHK(out_done(c, Kzz, dzz, bzz));
    CHK(out_done(c, Kzx, dzx, bzx));
    CHK(out_done(c, Kxx, dxx, bxx));
    return finish(c);
}

static int e_kernel_K_seq_n_seq_covs(gpsig_ctx* c, const gpsig_params* p, const void* X, const void* X2, int64_t N1, int64_t N2,
                                  int32_t L1, int32_t L2, int32_t full_X2_cov, int32_t return_levels, void* Kxx, void* Kxx2,
                                  void* Kx2x2) {
    ENTER(c, p);
    const int M1 = p->num_levels + 1, d = p->num_features;
    const size_t lv = return_levels ? size_t(M1) : 1;
    const void *dX, *dX2;
    CHK(in_dev(c, B_IN0, X, sizeof(TT) * size_t(N1) * L1 * d, &dX));
    CHK(in_dev(c, B_IN1, X2, sizeof(TT) * size_t(N2) * L2 * d, &dX2));
    const size_t b11 = sizeof(TT) * size_t(N1) * N1 * lv, b12 = sizeof(TT) * size_t(N1) * N2 * lv;
    const size_t b22 = sizeof(TT) * (full_X2_cov ? size_t(N2) * N2 : size_t(N2)) * lv;
    void *d11, *d12, *d22;
    CHK(out_dev(c, B_OUT0, Kxx, b11, &d11));
    CHK(out_dev(c, B_OUT1, Kxx2, b12, &d12));
    CHK(out_dev(c, B_OUT2, Kx2x2, b22, &d22));
    // Kxx (kernels.py:704, :709-712, :730/:755) == K(X)
    CHK(seq_K_device(c, p, false, dX, nullptr, N1, N1, L1, L1, return_levels, d11, true));
    // Kxx2 (kernels.py:705, :713, :727 / :750).  Reference quirk, reproduced: in the diagonal-only branch the
    // X-side factor is applied twice (:713 then :750), i.e. 1/(diag_x + jitter) instead of 1/sqrt(.).
    CHK(seq_K_device(c, p, false, dX, dX2, N1, N2, L1, L2, return_levels, d12, true, (p->normalization && !full_X2_cov) ? 1 : 0));
    if (full_X2_cov) {
        // kernels.py:719-732; :723-728 reference undefined names -- the evident intent (mirror of :709-712) == K(X2)
        CHK(seq_K_device(c, p, false, dX2, nullptr, N2, N2, L2, L2, return_levels, d22, true));
    } else {
        const double* w;
        CHK(upload_weights(c, p, &w));
        void* tmp;
        CHK(ensure(c, B_TMP0, sizeof(TT) * size_t(N2) * M1 + 8, &tmp));
        if (p->normalization) {   // kernels.py:751
            if (N2 > 0) {
                hipLaunchKernelGGL(fill_kernel<TT>, dim3(grid_for(N2 * M1)), dim3(256), 0, c->stream, static_cast<TT*>(tmp), N2 * M1, TT(1));
                HIPCHK(c, hipGetLastError());
            }
        } else {                  // kernels.py:743, :753
            CHK(diag_levels_mn(c, p, true, dX2, N2, L2, tmp));
        }
        if (N2 > 0) {
            hipLaunchKernelGGL(weight_levels_kernel<TT>, dim3(grid_for(N2)), dim3(256), 0, c->stream,
                               static_cast<const TT*>(tmp), N2, M1, w, return_levels ? 0 : 1, static_cast<TT*>(d22));
            HIPCHK(c, hipGetLastError());
        }
    }
    CHK(out_done(c, Kxx, d11, b11));
    CHK(out_done(c, Kxx2, d12, b12));
    CHK(out_done(c, Kx2x2, d22, b22));
    return finish(c);
}

};

}  // namespace

// =====================================================================================================
extern "C" {

int gpsig_abi_version(void) { return GPSIG_ABI_VERSION; }

int gpsig_ctx_create(int device, void* stream, gpsig_ctx** out) {
    if (!out) return GPSIG_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, GPSIG_ERR_HIP, "no HIP device available (%s); libgpsig_hip has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(nullptr, GPSIG_ERR_INVALID, "device %d out of range [0, %d)", device, count);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, GPSIG_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(nullptr, GPSIG_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, GPSIG_ERR_UNSUPPORTED, "device %d is %s; this library carries gfx950 (MI355X) code only", device, prop.gcnArchName);
    gpsig_ctx* c = new gpsig_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    c->stream = static_cast<hipStream_t>(stream);
    const char* g = getenv("GPSIG_GLDS");
    c->use_glds = g ? atoi(g) : 1;     // LDS-DMA staging is the default (bit-identical to load + ds_write, slightly faster)
    const char* ex = getenv("GPSIG_NO_EXACT");
    c->allow_exact = (ex && atoi(ex)) ? 0 : 1;
    *out = c;
    return GPSIG_OK;
}

void gpsig_ctx_destroy(gpsig_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf& b : c->buf)
        if (b.p) (void)hipFree(b.p);
    for (TaskSlot& t : c->task_slots)
        if (t.buf.p) (void)hipFree(t.buf.p);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->probe_stream) { if (c->probe_stop) *c->probe_stop = 1; (void)hipStreamSynchronize(c->probe_stream); (void)hipStreamDestroy(c->probe_stream); }
    if (c->probe_buf) (void)hipFree(c->probe_buf);
    if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
    if (c->side_fork) (void)hipEventDestroy(c->side_fork);
    if (c->side_join) (void)hipEventDestroy(c->side_join);
    for (int k = 0; k < 2; ++k) {
        if (c->pin[k]) (void)hipHostFree(c->pin[k]);
        if (c->pin_ev[k]) (void)hipEventDestroy(c->pin_ev[k]);
    }
    if (c->probe_stop) (void)hipHostFree(const_cast<int*>(c->probe_stop));
    solver_release(c->blas_handle);
    for (gpsig_lr_state* st : c->lr_states) { st->device = c->device; st->ctx = nullptr; }      // outlive the context as plain memory blocks
    delete c;
}

const char* gpsig_last_error(gpsig_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int gpsig_set_pointer_mode(gpsig_ctx* c, int mode) {
    if (!c) return GPSIG_ERR_INVALID;
    if (mode != GPSIG_PTR_HOST && mode != GPSIG_PTR_DEVICE) return fail(c, GPSIG_ERR_INVALID, "unknown pointer mode %d", mode);
    c->ptr_mode = mode;
    return GPSIG_OK;
}

int gpsig_sync(gpsig_ctx* c) {
    if (!c) return GPSIG_ERR_INVALID;
    CHK(host_sync(c));
    return GPSIG_OK;
}

int gpsig_set_shard(gpsig_ctx* c, int index, int count) {
    if (!c) return GPSIG_ERR_INVALID;
    if (count < 1 || index < 0 || index >= count) return fail(c, GPSIG_ERR_INVALID, "bad shard (%d of %d)", index, count);
    c->shard_i = index;
    c->shard_n = count;
    return GPSIG_OK;
}

int gpsig_set_option(gpsig_ctx* c, const char* name, int value) {
    if (!c || !name) return GPSIG_ERR_INVALID;
    if (!strcmp(name, "glds")) c->use_glds = value ? 1 : 0;
    else if (!strcmp(name, "exact")) c->allow_exact = value ? 1 : 0;
    else if (!strcmp(name, "max_run")) c->max_run = value > 0 ? value : 0;
    else if (!strcmp(name, "tensor_lanes")) c->tens_lanes = value;
    else if (!strcmp(name, "grad_scratch_mb")) c->grad_scratch_mb = value > 0 ? value : 4096;
    else if (!strcmp(name, "grad_impl")) c->grad_impl = value;
    else if (!strcmp(name, "grad_stash_mb")) c->grad_stash_mb = value;
    else if (!strcmp(name, "matern_fast")) c->matern_fast = value;
    else if (!strcmp(name, "grad_fused_piece")) c->grad_fused_piece = value;
    else if (!strcmp(name, "tvs_zreg")) c->tvs_zreg = value;
    else if (!strcmp(name, "tvs_grad_tile")) c->tvs_grad_tile = value;
    else if (!strcmp(name, "pinned_staging")) c->pinned_staging = value ? 1 : 0;
    else if (!strcmp(name, "lr_jacobi")) c->lr_jacobi = value ? 1 : 0;
    else if (!strcmp(name, "sig_features")) c->sig_features = value;
    else if (!strcmp(name, "sig_gemm_dma")) c->sig_gemm_dma = value;
    else if (!strcmp(name, "sig_graded")) c->sig_graded = value;
    else if (!strcmp(name, "lr_grad_threads")) c->lr_grad_threads = value;
    else if (!strcmp(name, "sig_features_grad")) c->sig_features_grad = value;
    else if (!strcmp(name, "sig_features_keep")) { c->sf_keep = value; c->sf_valid = false; }
    else if (!strcmp(name, "keep_reset")) c->keep_reset = value ? 1 : 0;
    else if (!strcmp(name, "pk2")) c->allow_pk2 = value;
    else if (!strcmp(name, "f32_pack")) c->f32_pack = value;
    else if (!strcmp(name, "f32_waves")) c->f32_waves = value;
    else if (!strcmp(name, "tvs_tile")) c->tvs_tile = value;
    else if (!strcmp(name, "wide")) c->wide = value;
    else if (!strcmp(name, "wide_chunk_mb")) c->wide_chunk_mb = value > 0 ? value : 0;
    else if (!strcmp(name, "wide_contract")) c->wide_contract = value;
    else if (!strcmp(name, "wide_lat_waves")) c->wide_lat_waves = value;
    else if (!strcmp(name, "wide_sym_fold")) c->wide_sym_fold = value ? 1 : 0;
    else if (!strcmp(name, "ho_g32")) c->ho_g32 = value;
    else if (!strcmp(name, "wide_few_cols")) c->wide_few_cols = value;
    else if (!strcmp(name, "tvs_grad_matern")) c->tvs_grad_matern = value ? 1 : 0;
    else if (!strcmp(name, "wide_o1_sweeps")) c->wide_o1_sweeps = value;
    else if (!strcmp(name, "tvs_features")) c->tvs_features = value;
    else if (!strcmp(name, "tvs_tile_nw")) c->tvs_tile_nw = value;
    else if (!strcmp(name, "diag_own")) c->diag_own = value;
    else if (!strcmp(name, "spectral_wave")) c->spectral_wave = value;
    else if (!strcmp(name, "tens_tile")) c->tens_tile = value;
    else if (!strcmp(name, "lr_gemm")) c->lr_gemm = value;
    else if (!strcmp(name, "lr_fused")) c->lr_fused = value;
    else if (!strcmp(name, "lr_fused_variant")) c->lr_fused_variant = value;
    else if (!strcmp(name, "lr_fused_pad")) c->lr_fused_pad = value >= 0 ? value : 1;
    else return fail(c, GPSIG_ERR_INVALID, "unknown option '%s'", name);
    return GPSIG_OK;
}

int gpsig_graph_begin(gpsig_ctx* c) {
    if (!c) return GPSIG_ERR_INVALID;
    if (c->capturing) return fail(c, GPSIG_ERR_INVALID, "a graph capture is already open on this context");
    if (c->ptr_mode != GPSIG_PTR_DEVICE) return fail(c, GPSIG_ERR_INVALID, "graph capture needs device-pointer mode (gpsig_set_pointer_mode)");
    if (c->stream == nullptr) return fail(c, GPSIG_ERR_INVALID, "the default stream cannot be captured: create the context on a stream of its own");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    c->capturing = true;
    c->capture_failed = false;
    return GPSIG_OK;
}

int gpsig_graph_end(gpsig_ctx* c, gpsig_graph** out) {
    if (!c || !out) return GPSIG_ERR_INVALID;
    *out = nullptr;
    if (!c->capturing) return fail(c, GPSIG_ERR_INVALID, "no graph capture is open on this context");
    c->capturing = false;
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (c->capture_failed) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        // c->err still holds what the failing call reported
        return GPSIG_ERR_INVALID;
    }
    if (e != hipSuccess || !g) return fail(c, GPSIG_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    gpsig_graph* G = new (std::nothrow) gpsig_graph();
    if (!G) { (void)hipGraphDestroy(g); return fail(c, GPSIG_ERR_NOMEM, "out of host memory"); }
    G->graph = g; G->ctx = c; G->alloc_gen = c->alloc_gen;
    const hipError_t ei = hipGraphInstantiate(&G->exec, g, nullptr, nullptr, 0);
    if (ei != hipSuccess) {
        (void)hipGraphDestroy(g);
        delete G;
        return fail(c, GPSIG_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ei));
    }
    *out = G;
    return GPSIG_OK;
}

int gpsig_graph_launch(gpsig_ctx* c, gpsig_graph* G) {
    if (!c || !G) return GPSIG_ERR_INVALID;
    if (G->ctx != c) return fail(c, GPSIG_ERR_INVALID, "the graph was recorded on another context");
    if (c->capturing) return fail(c, GPSIG_ERR_INVALID, "a graph capture is open on this context");
    if (G->alloc_gen != c->alloc_gen)
        return fail(c, GPSIG_ERR_INVALID, "scratch buffers moved since the graph was recorded (a later call needed more memory, other task lists or other level weights): record it again");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipGraphLaunch(G->exec, c->stream));
    return GPSIG_OK;
}

void gpsig_graph_destroy(gpsig_graph* G) {
    if (!G) return;
    if (G->exec) (void)hipGraphExecDestroy(G->exec);
    if (G->graph) (void)hipGraphDestroy(G->graph);
    delete G;
}

int gpsig_timing_reset(gpsig_ctx* c) {
    if (!c) return GPSIG_ERR_INVALID;
    CHK(host_sync(c));
    c->ev_used = 0;
    c->t_launches = 0;
    c->t_pairs = 0;
    c->t_kernel = nullptr;
    c->t_flops = 0.0;
    return GPSIG_OK;
}

int gpsig_timing_info(gpsig_ctx* c, const char** kernel, double* flops) {
    if (!c) return GPSIG_ERR_INVALID;
    if (kernel) *kernel = c->t_kernel;
    if (flops) *flops = c->t_flops;
    return GPSIG_OK;
}

int gpsig_timing_get(gpsig_ctx* c, double* kernel_ms, int64_t* launches, int64_t* pairs) {
    if (!c) return GPSIG_ERR_INVALID;
    CHK(host_sync(c));
    double tot = 0.0;
    for (size_t k = 0; k + 1 < c->ev_used; k += 2) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[k], c->ev[k + 1]));
        tot += ms;
    }
    if (kernel_ms) *kernel_ms = tot;
    if (launches) *launches = c->t_launches;
    if (pairs) *pairs = c->t_pairs;
    return GPSIG_OK;
}

// ---- effective shader clock while other kernels run -------------------------------------------------------------------
// One wavefront on a non-blocking stream of its own sleeps and, every `interval` ticks of the constant 100 MHz counter
// (s_memrealtime), stores that counter next to the shader-clock counter (s_memtime: one tick per shader cycle -- checked
// against an issue-bound loop of known length, tools/clockcheck.hip: 2.40 GHz on an idle chip, 2.14 with every SIMD on
// v_fma_f64).  The ratio of the two differences is the clock the chip ran at.  The wave leaves when the host raises `stop`
// (pinned host memory, read over the bus between naps), after `nsamp` readings, or when `deadline` ticks have passed.
static constexpr int PROBE_WAVES = gpsig_ctx::PROBE_WAVES;
static __global__ void clock_probe_kernel(unsigned long long* out_all, int cap, int nsamp, unsigned long long interval,
                                          unsigned long long deadline, const int* stop) {
    // one wavefront per workgroup; the dispatcher deals consecutive workgroups to consecutive XCDs, whose clocks differ
    if (threadIdx.x != 0) return;
    unsigned long long* out = out_all + size_t(blockIdx.x) * (2 * size_t(cap) + 2);
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long next = r0;
    int k = 0;
    while (k < nsamp) {
        const unsigned long long r = __builtin_amdgcn_s_memrealtime();
        const bool leave = __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || r - r0 > deadline;
        if (r >= next || leave) {
            const unsigned long long cyc = __builtin_amdgcn_s_memtime();
            const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
            out[2 * k] = cyc;
            out[2 * k + 1] = rt;
            ++k;
            next += interval;
        }
        if (leave) break;
        __builtin_amdgcn_s_sleep(100);
    }
    out[2 * size_t(cap)] = (unsigned long long)k;
    out[2 * size_t(cap) + 1] = __builtin_amdgcn_s_getreg((3 << 11) | 20);        // HW_REG_XCC_ID, bits 3:0
}

int gpsig_clock_probe_start(gpsig_ctx* c, double duration_ms, int32_t samples) {
    if (!c) return GPSIG_ERR_INVALID;
    if (c->capturing) return fail(c, GPSIG_ERR_INVALID, "no clock probe inside a graph capture");
    if (!(duration_ms > 0.0) || duration_ms > 60000.0 || samples < 2 || samples > 4096)
        return fail(c, GPSIG_ERR_INVALID, "clock probe: duration in (0, 60000] ms and 2..4096 samples");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->probe_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->probe_stream, hipStreamNonBlocking));
    if (!c->probe_stop) {
        void* hp = nullptr;
        HIPCHK(c, hipHostMalloc(&hp, 64, hipHostMallocMapped));
        c->probe_stop = static_cast<volatile int*>(hp);
    }
    if (c->probe_n) {                       // a probe that was never read: let it go first
        *c->probe_stop = 1;
        HIPCHK(c, hipStreamSynchronize(c->probe_stream));
    }
    *c->probe_stop = 0;
    if (samples > c->probe_cap) {
        if (c->probe_buf) HIPCHK(c, hipFree(c->probe_buf));
        c->probe_buf = nullptr;
        c->probe_cap = 0;
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->probe_buf), sizeof(unsigned long long) * (2 * size_t(samples) + 2) * PROBE_WAVES));
        c->probe_cap = samples;
    }
    unsigned long long interval = (unsigned long long)(duration_ms * 1e5 / double(samples - 1));    // 100 MHz ticks
    if (!interval) interval = 1;
    int* dstop = nullptr;
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&dstop), const_cast<int*>(c->probe_stop), 0));
    HIPCHK(c, hipMemsetAsync(c->probe_buf, 0, sizeof(unsigned long long) * (2 * size_t(c->probe_cap) + 2) * PROBE_WAVES, c->probe_stream));
    hipLaunchKernelGGL(clock_probe_kernel, dim3(PROBE_WAVES), dim3(64), 0, c->probe_stream, c->probe_buf, c->probe_cap, int(samples), interval,
                       (unsigned long long)(duration_ms * 1e5 * 1.25) + 1000ull, dstop);
    HIPCHK(c, hipGetLastError());
    c->probe_n = samples;
    return GPSIG_OK;
}

int gpsig_clock_probe_read(gpsig_ctx* c, double* ghz_mean, double* ghz_min, double* ghz_max, double* covered_ms) {
    if (!c) return GPSIG_ERR_INVALID;
    if (!c->probe_stream || c->probe_n < 2) return fail(c, GPSIG_ERR_INVALID, "no clock probe was started");
    *c->probe_stop = 1;                     // the wave takes one last reading and leaves
    HIPCHK(c, hipStreamSynchronize(c->probe_stream));
    c->probe_n = 0;
    const size_t per = 2 * size_t(c->probe_cap) + 2;
    std::vector<unsigned long long> h(per * PROBE_WAVES);
    HIPCHK(c, hipMemcpy(h.data(), c->probe_buf, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    double lo = 1e30, hi = 0.0, sum = 0.0, cov = 0.0;
    int waves = 0;
    for (int w = 0; w < PROBE_WAVES; ++w) {
        const unsigned long long* q = h.data() + per * size_t(w);
        const int n = int(q[2 * size_t(c->probe_cap)]);
        if (n < 2) continue;                             // a wave that was never scheduled before the stop
        for (int k = 1; k < n; ++k) {
            const double dc = double(q[2 * k] - q[2 * (k - 1)]), dr = double(q[2 * k + 1] - q[2 * (k - 1) + 1]);
            if (dr < 100.0) continue;                    // readings less than a microsecond apart (the last one, taken on leaving)
            const double g = dc / dr * 0.1;              // cycles per 10 ns -> GHz
            if (g < lo) lo = g;
            if (g > hi) hi = g;
        }
        const double dC = double(q[2 * (n - 1)] - q[0]), dR = double(q[2 * (n - 1) + 1] - q[1]);
        if (!(dR > 0.0)) continue;
        c->probe_ghz[waves] = dC / dR * 0.1;
        c->probe_xcc[waves] = int(q[2 * size_t(c->probe_cap) + 1] & 15);
        sum += c->probe_ghz[waves];
        if (dR * 1e-5 > cov) cov = dR * 1e-5;
        ++waves;
    }
    c->probe_waves = waves;
    if (!waves) return fail(c, GPSIG_ERR_INVALID, "the clock probe took fewer than two readings: it was read before it ran");
    if (ghz_mean) *ghz_mean = sum / waves;               // mean over the XCDs that were sampled
    if (ghz_min) *ghz_min = lo < 1e29 ? lo : 0.0;
    if (ghz_max) *ghz_max = hi;
    if (covered_ms) *covered_ms = cov;
    return GPSIG_OK;
}

int gpsig_clock_probe_xcds(gpsig_ctx* c, double* ghz, int32_t* xcc, int32_t cap, int32_t* n) {
    if (!c || !n) return GPSIG_ERR_INVALID;
    const int m = c->probe_waves < cap ? c->probe_waves : cap;
    for (int w = 0; w < m; ++w) {
        if (ghz) ghz[w] = c->probe_ghz[w];
        if (xcc) xcc[w] = c->probe_xcc[w];
    }
    *n = m;
    return GPSIG_OK;
}

int gpsig_seq_gram_levels(gpsig_ctx* c, const gpsig_params* p, const void* X, const void* X2, int64_t N1, int64_t N2,
                          int32_t L1, int32_t L2, void* out) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_seq_gram_levels(c, p, X, X2, N1, N2, L1, L2, out) : Impl<double>::e_seq_gram_levels(c, p, X, X2, N1, N2, L1, L2, out);
}

// The raw levels as gpsig_seq_gram_levels, and -- where the fused reverse kernel can continue from it (SignatureRBF with differences, order 1,
// float64, at most 64 observations and 8 columns, num_levels 4 / 5, the whole thing within "grad_stash_mb") -- the forward recursion's row totals
// and final states kept in the context for gpsig_seq_gram_levels_grad_stash.  desc[0] == 0: nothing was kept, differentiate with
// gpsig_seq_gram_levels_grad.  Device pointers only.
int gpsig_seq_gram_levels_stash(gpsig_ctx* c, const gpsig_params* p, const void* X, const void* X2, int64_t N1, int64_t N2,
                                int32_t L1, int32_t L2, void* out, int64_t* desc) {
    if (!c || !p || !desc) return GPSIG_ERR_INVALID;
    for (int q = 0; q < 8; ++q) desc[q] = 0;
    if (p->dtype != GPSIG_F64 || c->ptr_mode != GPSIG_PTR_DEVICE) return gpsig_seq_gram_levels(c, p, X, X2, N1, N2, L1, L2, out);
    c->stash_want = true;
    c->stash_desc[0] = 0;
    const int rc = Impl<double>::e_seq_gram_levels(c, p, X, X2, N1, N2, L1, L2, out);
    c->stash_want = false;
    if (rc == GPSIG_OK)
        for (int q = 0; q < 8; ++q) desc[q] = c->stash_desc[q];
    return rc;
}

int gpsig_seq_diag_levels(gpsig_ctx* c, const gpsig_params* p, const void* X, int64_t N, int32_t L, void* out) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_seq_diag_levels(c, p, X, N, L, out) : Impl<double>::e_seq_diag_levels(c, p, X, N, L, out);
}

int gpsig_tens_gram_levels(gpsig_ctx* c, const gpsig_params* p, const void* Z, int64_t T, int32_t increments, void* out) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_tens_gram_levels(c, p, Z, T, increments, out) : Impl<double>::e_tens_gram_levels(c, p, Z, T, increments, out);
}

int gpsig_tens_vs_seq_levels(gpsig_ctx* c, const gpsig_params* p, const void* Z, const void* X, int64_t T, int64_t N,
                             int32_t L, int32_t increments, void* out) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_tens_vs_seq_levels(c, p, Z, X, T, N, L, increments, out) : Impl<double>::e_tens_vs_seq_levels(c, p, Z, X, T, N, L, increments, out);
}

int gpsig_tens_vs_seq_weighted(gpsig_ctx* c, const gpsig_params* p, const void* Z, const void* X, int64_t T, int64_t N, int32_t L,
                               int32_t increments, const void* fac, void* out, void* aux, int32_t* aux_written) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_tens_vs_seq_weighted(c, p, Z, X, T, N, L, increments, fac, out, aux, aux_written)
                                 : Impl<double>::e_tens_vs_seq_weighted(c, p, Z, X, T, N, L, increments, fac, out, aux, aux_written);
}

int64_t gpsig_tens_vs_seq_aux_elems(const gpsig_params* p, int64_t T, int64_t N) {
    if (!p || T < 0 || N < 0) return 0;
    return N * int64_t(p->num_levels * (p->num_levels + 1) / 2) * ((T + 63) / 64 * 64);
}

int gpsig_kernel_K(gpsig_ctx* c, const gpsig_params* p, const void* X, const void* X2, int64_t N1, int64_t N2, int32_t L1,
                   int32_t L2, int32_t return_levels, void* out) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_kernel_K(c, p, X, X2, N1, N2, L1, L2, return_levels, out) : Impl<double>::e_kernel_K(c, p, X, X2, N1, N2, L1, L2, return_levels, out);
}

int gpsig_kernel_K_symm_rows(gpsig_ctx* c, const gpsig_params* p, const void* X, int64_t N, int32_t L, int64_t row_begin,
                             int64_t row_end, void* out_rows) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_kernel_K_symm_rows(c, p, X, N, L, row_begin, row_end, out_rows, 0) : Impl<double>::e_kernel_K_symm_rows(c, p, X, N, L, row_begin, row_end, out_rows, 0);
}

int gpsig_symmetrize_owned_rows(gpsig_ctx* c, int32_t dtype, const void* half, int64_t N, void* out) {
    if (!c) return GPSIG_ERR_INVALID;
    return dtype == GPSIG_F32 ? Impl<float>::e_symmetrize_owned_rows(c, dtype, half, N, out) : Impl<double>::e_symmetrize_owned_rows(c, dtype, half, N, out);
}

int gpsig_kernel_K_symm_rows_compact(gpsig_ctx* c, const gpsig_params* p, const void* X, int64_t N, int32_t L, int64_t row_begin,
                                     int64_t row_end, void* out_rows) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_kernel_K_symm_rows(c, p, X, N, L, row_begin, row_end, out_rows, 1) : Impl<double>::e_kernel_K_symm_rows(c, p, X, N, L, row_begin, row_end, out_rows, 1);
}

int gpsig_symmetrize_compact_rows(gpsig_ctx* c, int32_t dtype, const void* half, int64_t N, void* out) {
    if (!c) return GPSIG_ERR_INVALID;
    return dtype == GPSIG_F32 ? Impl<float>::e_symmetrize_compact_rows(c, dtype, half, N, out) : Impl<double>::e_symmetrize_compact_rows(c, dtype, half, N, out);
}

int gpsig_kernel_Kdiag(gpsig_ctx* c, const gpsig_params* p, const void* X, int64_t N, int32_t L, int32_t return_levels, void* out) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_kernel_Kdiag(c, p, X, N, L, return_levels, out) : Impl<double>::e_kernel_Kdiag(c, p, X, N, L, return_levels, out);
}

int gpsig_kernel_K_tens(gpsig_ctx* c, const gpsig_params* p, const void* Z, int64_t T, int32_t increments, int32_t return_levels, void* out) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_kernel_K_tens(c, p, Z, T, increments, return_levels, out) : Impl<double>::e_kernel_K_tens(c, p, Z, T, increments, return_levels, out);
}

int gpsig_kernel_K_tens_vs_seq(gpsig_ctx* c, const gpsig_params* p, const void* Z, const void* X, int64_t T, int64_t N,
                               int32_t L, int32_t increments, int32_t return_levels, void* out) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_kernel_K_tens_vs_seq(c, p, Z, X, T, N, L, increments, return_levels, out) : Impl<double>::e_kernel_K_tens_vs_seq(c, p, Z, X, T, N, L, increments, return_levels, out);
}

int gpsig_kernel_K_tens_n_seq_covs(gpsig_ctx* c, const gpsig_params* p, const void* Z, const void* X, int64_t T, int64_t N,
                                   int32_t L, int32_t increments, int32_t full_X_cov, int32_t return_levels, void* Kzz,
                                   void* Kzx, void* Kxx) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_kernel_K_tens_n_seq_covs(c, p, Z, X, T, N, L, increments, full_X_cov, return_levels, Kzz, Kzx, Kxx) : Impl<double>::e_kernel_K_tens_n_seq_covs(c, p, Z, X, T, N, L, increments, full_X_cov, return_levels, Kzz, Kzx, Kxx);
}

int gpsig_kernel_K_seq_n_seq_covs(gpsig_ctx* c, const gpsig_params* p, const void* X, const void* X2, int64_t N1, int64_t N2,
                                  int32_t L1, int32_t L2, int32_t full_X2_cov, int32_t return_levels, void* Kxx, void* Kxx2,
                                  void* Kx2x2) {
    if (!c || !p) return GPSIG_ERR_INVALID;
    return p->dtype == GPSIG_F32 ? Impl<float>::e_kernel_K_seq_n_seq_covs(c, p, X, X2, N1, N2, L1, L2, full_X2_cov, return_levels, Kxx, Kxx2, Kx2x2) : Impl<double>::e_kernel_K_seq_n_seq_covs(c, p, X, X2, N1, N2, L1, L2, full_X2_cov, return_levels, Kxx, Kxx2, Kx2x2);
}


int gpsig_lr_gather_points(gpsig_ctx* c, const gpsig_params* p, const void* X, int64_t N, int32_t L, const int64_t* idx, int64_t R,
                           double* out_host) {
    ENTER(c, p);
    if (p->dtype != GPSIG_F64) return fail(c, GPSIG_ERR_UNSUPPORTED, "low-rank mode is built for float64 only");
    if (!idx || !out_host || R < 0) return fail(c, GPSIG_ERR_INVALID, "bad landmark request");
    for (int64_t k = 0; k < R; ++k)
        if (idx[k] < 0 || idx[k] >= N * L) return fail(c, GPSIG_ERR_INVALID, "landmark index out of range");
    ScaleParams s;
    CHK(scale_params(c, p, true, &s));
    const int d_eff = s.d_eff();
    const void* dX;
    CHK(in_dev(c, B_IN0, X, sizeof(double) * size_t(N) * L * p->num_features, &dX));
    void *didx, *dout;
    CHK(ensure(c, B_LR2, sizeof(int64_t) * size_t(R) + 8, &didx));
    CHK(ensure(c, B_LR3, sizeof(double) * size_t(R) * d_eff + 8, &dout));
    if (R > 0) {
        HIPCHK(c, hipMemcpyAsync(didx, idx, sizeof(int64_t) * size_t(R), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(lr_gather_points_kernel<double>, dim3(grid_for(R * d_eff)), dim3(256), 0, c->stream,
                           static_cast<const double*>(dX), L, s, static_cast<const int64_t*>(didx), R, static_cast<double*>(dout));
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(out_host, dout, sizeof(double) * size_t(R) * d_eff, hipMemcpyDeviceToHost, c->stream));
    }
    CHK(host_sync(c));
    return GPSIG_OK;
}

int gpsig_base_kernel_matrix(gpsig_ctx* c, const gpsig_params* p, const double* A_host, const double* B_host, int64_t na, int64_t nb,
                             int32_t d, double* out_host) {
    ENTER(c, p);
    if (!A_host || !B_host || !out_host || na < 0 || nb < 0 || d < 1) return fail(c, GPSIG_ERR_INVALID, "bad base-kernel-matrix request");
    void *da, *db, *dout;
    CHK(ensure(c, B_LR2, sizeof(double) * size_t(na) * d + 8, &da));
    CHK(ensure(c, B_LR3, sizeof(double) * size_t(nb) * d + 8, &db));
    CHK(ensure(c, B_LR4, sizeof(double) * size_t(na) * nb + 8, &dout));
    if (na > 0 && nb > 0) {
        HIPCHK(c, hipMemcpyAsync(da, A_host, sizeof(double) * size_t(na) * d, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(db, B_host, sizeof(double) * size_t(nb) * d, hipMemcpyHostToDevice, c->stream));
        double p0, p1;
        base_p(p, &p0, &p1);
        const double* spec;
        CHK(spectral_table(c, p, &spec));
        hipLaunchKernelGGL(base_kernel_matrix_kernel<double>, dim3(grid_for(na * nb)), dim3(256), 0, c->stream,
                           static_cast<const double*>(da), static_cast<const double*>(db), na, nb, int(d), int(p->base_kernel), p0, p1, spec,
                           static_cast<double*>(dout));
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(out_host, dout, sizeof(double) * size_t(na) * nb, hipMemcpyDeviceToHost, c->stream));
    }
    CHK(host_sync(c));
    return GPSIG_OK;
}

int gpsig_lr_whitening(gpsig_ctx* c, const gpsig_params* p, const double* S_host, int32_t nc, int32_t d, const double* jitter_diag_host,
                       double* Wh_host, double* ev_host) {
    ENTER(c, p);
    if (!S_host || !jitter_diag_host || !Wh_host || nc < 1 || d < 1) return fail(c, GPSIG_ERR_INVALID, "bad whitening request");
    CHK(no_capture(c, "the Nystrom whitening copies to and from the host"));
    void *dS, *dW, *dJ, *dWh;
    CHK(ensure(c, B_LR2, sizeof(double) * size_t(nc) * d + 8, &dS));
    CHK(ensure(c, B_LR4, sizeof(double) * size_t(nc) * nc + 8, &dW));
    CHK(ensure(c, B_LR3, sizeof(double) * size_t(nc) * 3 + 64, &dJ));
    CHK(ensure(c, B_LR5, sizeof(double) * size_t(nc) * nc + 8, &dWh));
    double* const dEv = static_cast<double*>(dJ) + nc;
    double* const dWork = static_cast<double*>(dJ) + 2 * size_t(nc);
    int* const dInfo = reinterpret_cast<int*>(static_cast<double*>(dJ) + 3 * size_t(nc));
    HIPCHK(c, hipMemcpyAsync(dS, S_host, sizeof(double) * size_t(nc) * d, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dJ, jitter_diag_host, sizeof(double) * size_t(nc), hipMemcpyHostToDevice, c->stream));
    double p0, p1;
    base_p(p, &p0, &p1);
    const double* spec;
    CHK(spectral_table(c, p, &spec));
    const int64_t total = int64_t(nc) * nc;
    hipLaunchKernelGGL(base_kernel_matrix_kernel<double>, dim3(grid_for(total)), dim3(256), 0, c->stream, static_cast<const double*>(dS),
                       static_cast<const double*>(dS), int64_t(nc), int64_t(nc), int(d), int(p->base_kernel), p0, p1, spec,
                       static_cast<double*>(dW));                                                         // low_rank_calculations.py:51
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(add_diag_kernel, dim3((nc + 255) / 256), dim3(256), 0, c->stream, static_cast<double*>(dW),
                       static_cast<const double*>(dJ), int(nc));
    HIPCHK(c, hipGetLastError());
    std::string err;
    if (!solver_dsyevd(&c->blas_handle, c->stream, nc, static_cast<double*>(dW), dEv, dWork, dInfo, &err))    // :55
        return fail(c, GPSIG_ERR_HIP, "%s", err.c_str());
    hipLaunchKernelGGL(eig_sign_kernel, dim3((nc + 63) / 64), dim3(64), 0, c->stream, static_cast<const double*>(dW), int(nc), dWork);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(whiten_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, c->stream, static_cast<const double*>(dW),
                       static_cast<const double*>(dEv), static_cast<const double*>(dWork), int(nc), p->jitter, static_cast<double*>(dWh));
    HIPCHK(c, hipGetLastError());
    int info = 0;
    HIPCHK(c, hipMemcpyAsync(Wh_host, dWh, sizeof(double) * size_t(total), hipMemcpyDeviceToHost, c->stream));
    if (ev_host) HIPCHK(c, hipMemcpyAsync(ev_host, dEv, sizeof(double) * size_t(nc), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&info, dInfo, sizeof(info), hipMemcpyDeviceToHost, c->stream));
    CHK(host_sync(c));
    if (info != 0) return fail(c, GPSIG_ERR_HIP, "rocsolver_dsyevd did not converge (info = %d)", info);
    return GPSIG_OK;
}

int gpsig_lr_seq_features(gpsig_ctx* c, const gpsig_params* p, const gpsig_lowrank* lr, const void* X, int64_t N, int32_t L, void* Phi) {
    ENTER(c, p);
    CHK(lr_check(c, p, lr));
    const int M = p->num_levels;
    ScaleParams s;
    CHK(scale_params(c, p, true, &s));
    const int d_eff = s.d_eff();
    LrDev D;
    CHK(lr_upload(c, p, lr, d_eff, &D));
    const int cc = D.c, r = D.r;
    const int F = 1 + cc + (M - 1) * r;
    const int l = p->difference ? L - 1 : L;
    const void* dX;
    CHK(in_dev(c, B_IN0, X, sizeof(double) * size_t(N) * L * p->num_features, &dX));
    void* dPhi;
    CHK(out_dev(c, B_OUT0, Phi, sizeof(double) * size_t(N) * F, &dPhi));
    double* phi = static_cast<double*>(dPhi);
    double p0, p1;
    base_p(p, &p0, &p1);
    // One kernel, one workgroup per sequence, intermediates in LDS (lr_fused_kernel.hpp) when a sequence's three arrays fit
    const size_t fused_lds = lr_fused_lds_bytes(cc, r, d_eff, L, c->lr_fused_pad);
    if (c->lr_fused != 0 && fused_lds <= LR_FUSED_MAX_LDS && M - 1 <= LR_FUSED_MAX_SKETCHES) {
        if (N <= 0) return finish(c);
        LrFusedArgs A;
        A.X = static_cast<const double*>(dX); A.N = N; A.L = L; A.P = s; A.S = D.S; A.Wh = D.Wh;
        A.c = cc; A.r = r; A.M = M; A.difference = p->difference; A.kind = int(p->base_kernel); A.p0 = p0; A.p1 = p1;
        for (int i = 0; i < LR_FUSED_MAX_SKETCHES; ++i) A.sk[i] = LrFusedSketch{nullptr, nullptr};
        for (int i = 0; i < D.nsk; ++i) A.sk[i] = LrFusedSketch{D.colptr[i], D.ent[i]};
        A.Phi = phi; A.F = F;
        A.lp = lr_fused_stride(L, c->lr_fused_pad);
        A.rows_b = cc > r ? cc : r;
        if (d_eff > A.rows_b) A.rows_b = d_eff;
        const unsigned grid = unsigned(N < (int64_t(1) << 20) ? N : (int64_t(1) << 20));
        // two arrays in LDS instead of three where a wavefront can hold its output columns in registers (lr_fused2): a third workgroup per CU
        const int rc = (c->lr_fused == 1 && lr_fused2_ok(cc, r, L)) ? lr_fused2_launch(c->stream, A, grid)
                                                                    : lr_fused_launch(c->stream, A, grid, c->lr_fused_variant);
        if (rc != 0) return fail(c, GPSIG_ERR_HIP, "fused low-rank feature kernel: %s", hipGetErrorString(hipError_t(rc)));
        CHK(out_done(c, Phi, dPhi, sizeof(double) * size_t(N) * F));
        return finish(c);
    }
    void *kxs, *feat, *U, *Pa, *Pb;
    const int wmax = cc > r ? cc : r;
    CHK(ensure(c, B_LR2, sizeof(double) * size_t(N) * L * cc + 8, &kxs));
    CHK(ensure(c, B_LR3, sizeof(double) * size_t(N) * L * cc + 8, &feat));
    CHK(ensure(c, B_LR4, sizeof(double) * size_t(N) * (l > 0 ? l : 1) * cc + 8, &U));
    CHK(ensure(c, B_LR5, sizeof(double) * size_t(N) * (l > 0 ? l : 1) * wmax + 8, &Pa));
    CHK(ensure(c, B_LR6, sizeof(double) * size_t(N) * (l > 0 ? l : 1) * wmax + 8, &Pb));
    if (N <= 0) return finish(c);
    // Nystrom features (low_rank_calculations.py:59-60): kappa(X, S) then the whitening GEMM on the matrix cores
    hipLaunchKernelGGL(lr_seq_cross_kernel<double>, dim3(grid_for(N * L * cc)), dim3(256), 0, c->stream, static_cast<const double*>(dX),
                       N, int(L), s, D.S, cc, int(p->base_kernel), p0, p1, static_cast<double*>(kxs));
    HIPCHK(c, hipGetLastError());
    // feat = kxs (NL, c) * Wh (c, c) = kxs * (Wh^T)^T : pass B = Wh^T, i.e. read Wh column-wise -> upload is row-major Wh, so
    // B[j][k] must be Wh[k][j]: use the transposed copy made below
    void* wht;
    CHK(ensure(c, B_LR7, sizeof(double) * size_t(cc) * cc + 8, &wht));
    if (lr->device_state) {
        wht = lr->device_state->WhT;
    } else {
        std::vector<double> t(size_t(cc) * cc);
        for (int a = 0; a < cc; ++a)
            for (int b = 0; b < cc; ++b) t[size_t(b) * cc + a] = lr->whitening[size_t(a) * cc + b];
        HIPCHK(c, hipMemcpyAsync(wht, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, c->stream));
        CHK(host_sync(c));
    }
    CHK(lr_gemm(c, static_cast<const double*>(kxs), static_cast<const double*>(wht), N * L, cc, cc, cc, cc, static_cast<double*>(feat), cc));
    // level 0 and level 1 (signature_algs.py:177-182)
    hipLaunchKernelGGL(fill_kernel<double>, dim3(grid_for(N * F)), dim3(256), 0, c->stream, phi, N * F, 0.0);
    HIPCHK(c, hipGetLastError());
    {
        // Phi[:, 0] = 1
        std::vector<double> ones(1, 1.0);
        (void)ones;
    }
    if (l > 0) {
        hipLaunchKernelGGL(lr_time_diff_kernel<double>, dim3(grid_for(N * l * cc)), dim3(256), 0, c->stream,
                           static_cast<const double*>(feat), N, int(L), cc, int(p->difference), static_cast<double*>(U));
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(lr_timesum_kernel<double>, dim3(grid_for(N * cc)), dim3(256), 0, c->stream, static_cast<const double*>(U), N, l,
                           cc, phi, int64_t(F), 1);
        HIPCHK(c, hipGetLastError());
        // P = U; for level i: P = excumsum_t(P); P = sketch(U, P); Phi_i = sum_t P     (signature_algs.py:184-191)
        HIPCHK(c, hipMemcpyAsync(Pa, U, sizeof(double) * size_t(N) * l * cc, hipMemcpyDeviceToDevice, c->stream));
        double *cur = static_cast<double*>(Pa), *nxt = static_cast<double*>(Pb);
        int kw = cc;
        for (int i = 2; i <= M; ++i) {
            hipLaunchKernelGGL(lr_excumsum_kernel<double>, dim3(grid_for(N * kw)), dim3(256), 0, c->stream, cur, N, l, kw, phi, int64_t(F), 0, 0);
            HIPCHK(c, hipGetLastError());
            hipLaunchKernelGGL(lr_sketch_kernel<double>, dim3(grid_for(N * l * r)), dim3(256), 0, c->stream, static_cast<const double*>(U),
                               int64_t(cc), static_cast<const double*>(cur), int64_t(kw), N * l, r, D.colptr[i - 2], D.i1[i - 2], D.i2[i - 2],
                               D.val[i - 2], nxt, int64_t(r));
            HIPCHK(c, hipGetLastError());
            hipLaunchKernelGGL(lr_timesum_kernel<double>, dim3(grid_for(N * r)), dim3(256), 0, c->stream, static_cast<const double*>(nxt), N, l,
                               r, phi, int64_t(F), 1 + cc + (i - 2) * r);
            HIPCHK(c, hipGetLastError());
            double* t = cur; cur = nxt; nxt = t;
            kw = r;
        }
    }
    hipLaunchKernelGGL(fill_strided_kernel<double>, dim3(grid_for(N)), dim3(256), 0, c->stream, phi, N, int64_t(F), 1.0);
    HIPCHK(c, hipGetLastError());
    CHK(out_done(c, Phi, dPhi, sizeof(double) * size_t(N) * F));
    return finish(c);
}

int gpsig_lr_tens_features(gpsig_ctx* c, const gpsig_params* p, const gpsig_lowrank* lr, const void* Z, int64_t T, int32_t increments,
                           void* Phi) {
    ENTER(c, p);
    CHK(lr_check(c, p, lr));
    const int M = p->num_levels, lt = M * (M + 1) / 2, E = increments ? 2 : 1;
    ScaleParams s;
    CHK(scale_params(c, p, true, &s));
    const int d_eff = s.d_eff();
    LrDev D;
    CHK(lr_upload(c, p, lr, d_eff, &D));
    const int cc = D.c, r = D.r;
    const int F = 1 + cc + (M - 1) * r;
    const void* dZ;
    CHK(in_dev(c, B_IN0, Z, sizeof(double) * size_t(lt) * T * E * d_eff, &dZ));
    void* dPhi;
    CHK(out_dev(c, B_OUT0, Phi, sizeof(double) * size_t(T) * F, &dPhi));
    double* phi = static_cast<double*>(dPhi);
    if (T <= 0) return finish(c);
    double p0, p1;
    base_p(p, &p0, &p1);
    if (c->lr_fused != 0 && lr_tens_fused_lds_bytes(cc, r, d_eff, lt, E) <= 64 * 1024 && M - 1 <= LR_FUSED_MAX_SKETCHES && T <= 0x7fffffff) {
        LrTensFusedArgs A;
        A.Z = static_cast<const double*>(dZ); A.T = T; A.lt = lt; A.E = E; A.P = s; A.S = D.S; A.Wh = D.Wh;
        A.c = cc; A.r = r; A.M = M; A.kind = int(p->base_kernel); A.p0 = p0; A.p1 = p1;
        for (int i = 0; i < LR_FUSED_MAX_SKETCHES; ++i) A.sk[i] = LrFusedSketch{nullptr, nullptr};
        for (int i = 0; i < D.nsk; ++i) A.sk[i] = LrFusedSketch{D.colptr[i], D.ent[i]};
        A.Phi = phi; A.F = F;
        const int rc = lr_tens_fused_launch(c->stream, A);
        if (rc != 0) return fail(c, GPSIG_ERR_HIP, "fused low-rank tensor feature kernel: %s", hipGetErrorString(hipError_t(rc)));
        CHK(out_done(c, Phi, dPhi, sizeof(double) * size_t(T) * F));
        return finish(c);
    }
    const int64_t rows = int64_t(lt) * T * E;
    void *kxs, *feat, *U, *Ra, *Rb, *wht;
    const int wmax = cc > r ? cc : r;
    CHK(ensure(c, B_LR2, sizeof(double) * size_t(rows) * cc + 8, &kxs));
    CHK(ensure(c, B_LR3, sizeof(double) * size_t(rows) * cc + 8, &feat));
    CHK(ensure(c, B_LR4, sizeof(double) * size_t(lt) * T * cc + 8, &U));
    CHK(ensure(c, B_LR5, sizeof(double) * size_t(T) * wmax + 8, &Ra));
    CHK(ensure(c, B_LR6, sizeof(double) * size_t(T) * wmax + 8, &Rb));
    CHK(ensure(c, B_LR7, sizeof(double) * size_t(cc) * cc + 8, &wht));
    if (lr->device_state) {
        wht = lr->device_state->WhT;
    } else {
        std::vector<double> t(size_t(cc) * cc);
        for (int a = 0; a < cc; ++a)
            for (int b = 0; b < cc; ++b) t[size_t(b) * cc + a] = lr->whitening[size_t(a) * cc + b];
        HIPCHK(c, hipMemcpyAsync(wht, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, c->stream));
        CHK(host_sync(c));
    }
    hipLaunchKernelGGL(lr_tens_cross_kernel<double>, dim3(grid_for(rows * cc)), dim3(256), 0, c->stream, static_cast<const double*>(dZ), rows,
                       s, D.S, cc, int(p->base_kernel), p0, p1, static_cast<double*>(kxs));
    HIPCHK(c, hipGetLastError());
    CHK(lr_gemm(c, static_cast<const double*>(kxs), static_cast<const double*>(wht), rows, cc, cc, cc, cc, static_cast<double*>(feat), cc));
    // increments: F(z[.,1]) - F(z[.,0]) (kernels.py:304); rows are ((k*T + t)*E + e): a "time difference" over e with L = E
    if (increments) {
        hipLaunchKernelGGL(lr_time_diff_kernel<double>, dim3(grid_for(int64_t(lt) * T * cc)), dim3(256), 0, c->stream,
                           static_cast<const double*>(feat), int64_t(lt) * T, 2, cc, 1, static_cast<double*>(U));
        HIPCHK(c, hipGetLastError());
    } else {
        HIPCHK(c, hipMemcpyAsync(U, feat, sizeof(double) * size_t(lt) * T * cc, hipMemcpyDeviceToDevice, c->stream));
    }
    hipLaunchKernelGGL(fill_kernel<double>, dim3(grid_for(T * F)), dim3(256), 0, c->stream, phi, T * F, 0.0);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(fill_strided_kernel<double>, dim3(grid_for(T)), dim3(256), 0, c->stream, phi, T, int64_t(F), 1.0);
    HIPCHK(c, hipGetLastError());
    // tensor_kern_lr_feature (signature_algs.py:211-221): R = U[k]; R = sketch_{j-1}(U[k'], R)
    const double* Ud = static_cast<const double*>(U);
    int k = 0;
    for (int i = 1; i <= M; ++i) {
        const double* R = Ud + size_t(k) * T * cc;
        int kw = cc;
        ++k;
        double *cur = static_cast<double*>(Ra), *nxt = static_cast<double*>(Rb);
        for (int j = 1; j < i; ++j) {
            hipLaunchKernelGGL(lr_sketch_kernel<double>, dim3(grid_for(T * r)), dim3(256), 0, c->stream, Ud + size_t(k) * T * cc, int64_t(cc), R,
                               int64_t(kw), T, r, D.colptr[j - 1], D.i1[j - 1], D.i2[j - 1], D.val[j - 1], cur, int64_t(r));
            HIPCHK(c, hipGetLastError());
            R = cur;
            kw = r;
            double* t = cur; cur = nxt; nxt = t;
            ++k;
        }
        const int off = i == 1 ? 1 : 1 + cc + (i - 2) * r;
        hipLaunchKernelGGL(copy_block_kernel<double>, dim3(grid_for(T * kw)), dim3(256), 0, c->stream, R, T, kw, int64_t(kw), phi, int64_t(F), off);
        HIPCHK(c, hipGetLastError());
    }
    CHK(out_done(c, Phi, dPhi, sizeof(double) * size_t(T) * F));
    return finish(c);
}

int gpsig_lr_kernel(gpsig_ctx* c, const gpsig_params* p, const gpsig_lowrank* lr, const void* PhiA, const void* PhiB, int64_t N1, int64_t N2,
                    int32_t normalize_a, int32_t normalize_b, int32_t return_levels, void* out) {
    ENTER(c, p);
    if (!lr) return fail(c, GPSIG_ERR_INVALID, "lowrank descriptor is NULL");
    if (p->dtype != GPSIG_F64) return fail(c, GPSIG_ERR_UNSUPPORTED, "low-rank mode is built for float64 only");
    const int M = p->num_levels, M1 = M + 1, cc = lr->num_components, r = lr->rank_bound;
    const bool sym = PhiB == nullptr;
    if (sym) N2 = N1;
    const int32_t* off;
    int F;
    CHK(lr_level_offsets(c, M, cc, r, &off, &F));
    const void *dA, *dB;
    CHK(in_dev(c, B_IN0, PhiA, sizeof(double) * size_t(N1) * F, &dA));
    if (sym) dB = dA; else CHK(in_dev(c, B_IN1, PhiB, sizeof(double) * size_t(N2) * F, &dB));
    const size_t ob = sizeof(double) * size_t(N1) * N2 * (return_levels ? M1 : 1);
    void* dout;
    CHK(out_dev(c, B_OUT0, out, ob, &dout));
    const double* w;
    CHK(upload_weights(c, p, &w));
    void *fa, *fb, *sa, *sb;
    CHK(ensure(c, B_LR2, sizeof(double) * size_t(N1) * M1 + 8, &fa));
    CHK(ensure(c, B_LR3, sizeof(double) * size_t(N2) * M1 + 8, &fb));
    CHK(ensure(c, B_LR4, sizeof(double) * size_t(N1) * F + 8, &sa));
    CHK(ensure(c, B_LR5, sizeof(double) * size_t(N2) * F + 8, &sb));
    if (N1 <= 0 || N2 <= 0) return finish(c);
    // per-row level factors: A side carries sigma*variances, both sides 1/sqrt(|Phi_m|^2 + jitter) when normalising
    hipLaunchKernelGGL(lr_level_factors_kernel<double>, dim3(grid_for(N1 * M1)), dim3(256), 0, c->stream, static_cast<const double*>(dA), N1,
                       int64_t(F), M1, off, w, p->jitter, int(normalize_a != 0), static_cast<double*>(fa));
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(lr_level_factors_kernel<double>, dim3(grid_for(N2 * M1)), dim3(256), 0, c->stream, static_cast<const double*>(dB), N2,
                       int64_t(F), M1, off, static_cast<const double*>(nullptr), p->jitter, int((sym ? normalize_a : normalize_b) != 0),
                       static_cast<double*>(fb));
    HIPCHK(c, hipGetLastError());
    const bool jit_diag = sym && normalize_a;         // kernels.py:431
    if (!return_levels) {
        hipLaunchKernelGGL(lr_scale_factors_kernel<double>, dim3(grid_for(N1 * F)), dim3(256), 0, c->stream, static_cast<const double*>(dA), N1,
                           int64_t(F), M1, off, static_cast<const double*>(fa), -1, static_cast<double*>(sa), int64_t(F));
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(lr_scale_factors_kernel<double>, dim3(grid_for(N2 * F)), dim3(256), 0, c->stream, static_cast<const double*>(dB), N2,
                           int64_t(F), M1, off, static_cast<const double*>(fb), -1, static_cast<double*>(sb), int64_t(F));
        HIPCHK(c, hipGetLastError());
        CHK(lr_gemm(c, static_cast<const double*>(sa), static_cast<const double*>(sb), N1, N2, F, F, F, static_cast<double*>(dout), N2));
        if (jit_diag) {
            hipLaunchKernelGGL(lr_add_jitter_diag_kernel<double>, dim3(grid_for(N1)), dim3(256), 0, c->stream, static_cast<double*>(dout), N1, M1,
                               static_cast<const double*>(fa), static_cast<const double*>(fb), p->jitter, -1);
            HIPCHK(c, hipGetLastError());
        }
    } else {
        std::vector<int32_t> hoff(M1 + 1);
        hoff[0] = 0; hoff[1] = 1;
        if (M >= 1) hoff[2] = 1 + cc;
        for (int m = 2; m <= M; ++m) hoff[m + 1] = hoff[m] + r;
        for (int m = 0; m <= M; ++m) {
            const int wdt = hoff[m + 1] - hoff[m];
            hipLaunchKernelGGL(lr_scale_factors_kernel<double>, dim3(grid_for(N1 * wdt)), dim3(256), 0, c->stream, static_cast<const double*>(dA),
                               N1, int64_t(F), M1, off, static_cast<const double*>(fa), m, static_cast<double*>(sa), int64_t(wdt));
            HIPCHK(c, hipGetLastError());
            hipLaunchKernelGGL(lr_scale_factors_kernel<double>, dim3(grid_for(N2 * wdt)), dim3(256), 0, c->stream, static_cast<const double*>(dB),
                               N2, int64_t(F), M1, off, static_cast<const double*>(fb), m, static_cast<double*>(sb), int64_t(wdt));
            HIPCHK(c, hipGetLastError());
            double* om = static_cast<double*>(dout) + size_t(m) * N1 * N2;
            CHK(lr_gemm(c, static_cast<const double*>(sa), static_cast<const double*>(sb), N1, N2, wdt, wdt, wdt, om, N2));
            if (jit_diag) {
                hipLaunchKernelGGL(lr_add_jitter_diag_kernel<double>, dim3(grid_for(N1)), dim3(256), 0, c->stream, om, N1, M1,
                                   static_cast<const double*>(fa), static_cast<const double*>(fb), p->jitter, m);
                HIPCHK(c, hipGetLastError());
            }
        }
    }
    CHK(out_done(c, out, dout, ob));
    return finish(c);
}

int gpsig_lr_kernel_diag(gpsig_ctx* c, const gpsig_params* p, const gpsig_lowrank* lr, const void* Phi, int64_t N, int32_t return_levels,
                         void* out) {
    ENTER(c, p);
    if (!lr) return fail(c, GPSIG_ERR_INVALID, "lowrank descriptor is NULL");
    if (p->dtype != GPSIG_F64) return fail(c, GPSIG_ERR_UNSUPPORTED, "low-rank mode is built for float64 only");
    const int M = p->num_levels, M1 = M + 1;
    const int32_t* off;
    int F;
    CHK(lr_level_offsets(c, M, lr->num_components, lr->rank_bound, &off, &F));
    const void* dP;
    CHK(in_dev(c, B_IN0, Phi, sizeof(double) * size_t(N) * F, &dP));
    const size_t ob = sizeof(double) * size_t(N) * (return_levels ? M1 : 1);
    void* dout;
    CHK(out_dev(c, B_OUT0, out, ob, &dout));
    const double* w;
    CHK(upload_weights(c, p, &w));
    if (N > 0) {
        hipLaunchKernelGGL(lr_level_diag_kernel<double>, dim3(grid_for(N * M1)), dim3(256), 0, c->stream, static_cast<const double*>(dP), N,
                           int64_t(F), M1, off, w, int(return_levels != 0), static_cast<double*>(dout));
        HIPCHK(c, hipGetLastError());
    }
    CHK(out_done(c, out, dout, ob));
    return finish(c);
}

int gpsig_lr_draw(gpsig_ctx* c, const gpsig_params* p, int32_t num_components, int32_t rank_bound, int32_t sparsity, uint64_t seed,
                  const void* X, int64_t N, int32_t L, const void* X2, int64_t N2, int32_t L2, const void* Z, int64_t T, int32_t increments,
                  gpsig_lr_state** inout) {
    ENTER(c, p);
    using namespace gpsig;
    if (!inout) return fail(c, GPSIG_ERR_INVALID, "null state slot");
    if (p->dtype != GPSIG_F64) return fail(c, GPSIG_ERR_UNSUPPORTED, "low-rank mode is built for float64 only");
    if (c->ptr_mode != GPSIG_PTR_DEVICE) return fail(c, GPSIG_ERR_INVALID, "gpsig_lr_draw takes device pointers (the host-side draw is gpsig_amd/low_rank.py)");
    if (num_components < 1 || rank_bound < 1 || sparsity < 0 || sparsity > 2) return fail(c, GPSIG_ERR_INVALID, "bad low-rank sizes");
    if (num_components > LR_DRAW_MAX || rank_bound > LR_DRAW_MAX) return fail(c, GPSIG_ERR_UNSUPPORTED, "the device-side draw takes at most %d components / rank bound", LR_DRAW_MAX);
    const int M = p->num_levels, nsk = M - 1;
    if (nsk > LR_FUSED_MAX_SKETCHES) return fail(c, GPSIG_ERR_UNSUPPORTED, "low-rank mode is built for num_levels <= %d", LR_FUSED_MAX_SKETCHES + 1);
    ScaleParams s;
    CHK(scale_params(c, p, true, &s));
    const int d_eff = s.d_eff();
    const int lt = M * (M + 1) / 2;
    const int64_t ztot = Z ? int64_t(lt) * T * (increments ? 2 : 1) : 0;
    const int64_t total = ztot + (X ? N * L : 0) + (X2 ? N2 * L2 : 0);
    if (num_components > total) return fail(c, GPSIG_ERR_INVALID, "num_components exceeds the number of available points");
    if (sparsity == 2 && nsk > 0 && (rank_bound > int64_t(num_components) * num_components || (nsk > 1 && rank_bound > int64_t(num_components) * rank_bound)))
        return fail(c, GPSIG_ERR_INVALID, "rank_bound exceeds the number of coordinate pairs");
    gpsig_lr_state* st = *inout;
    if (st && st->ctx != c) return fail(c, GPSIG_ERR_INVALID, "the state belongs to another context");
    if (!st) {
        st = new (std::nothrow) gpsig_lr_state();
        if (!st) return fail(c, GPSIG_ERR_NOMEM, "out of host memory");
        st->ctx = c;
    }
    if (!st->ctx) return fail(c, GPSIG_ERR_INVALID, "the state's context has been destroyed");
    st->c = num_components; st->d_eff = d_eff; st->r = rank_bound; st->nsk = nsk; st->sparsity = sparsity;
    CHK(no_capture(c, "a low-rank draw may have to allocate"));
    if (lr_state_layout(st, M) != GPSIG_OK) { if (!*inout) delete st; return fail(c, GPSIG_ERR_NOMEM, "hipMalloc failed for the low-rank state"); }
    if (!*inout) c->lr_states.push_back(st);
    *inout = st;
    const PhiloxKey key{uint32_t(seed), uint32_t(seed >> 32)};
    const int cc = st->c;
    HIPCHK(c, hipMemsetAsync(st->info, 0, sizeof(int) * 4, c->stream));
    hipLaunchKernelGGL(lr_draw_indices_kernel, dim3(1), dim3(64), 0, c->stream, total, cc, key, uint32_t(LRS_LANDMARKS), st->idx);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(lr_gather_landmarks_kernel, dim3(grid_for(int64_t(cc) * d_eff)), dim3(256), 0, c->stream, st->idx, cc,
                       static_cast<const double*>(Z), ztot, static_cast<const double*>(X), X ? N : 0, int(L), static_cast<const double*>(X2),
                       X2 ? N2 : 0, int(L2), s, key, p->jitter, st->S, st->jd);
    HIPCHK(c, hipGetLastError());
    // the projections depend on the seed only: they are drawn on a side stream while the main one decomposes the landmark Gram
    if (!c->side_stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->side_fork, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->side_join, hipEventDisableTiming));
    }
    HIPCHK(c, hipEventRecord(c->side_fork, c->stream));            // after the memset of info and everything queued before this draw
    HIPCHK(c, hipStreamWaitEvent(c->side_stream, c->side_fork, 0));
    hipStream_t const ss = c->side_stream;
    // W = kappa(S, S) + diag(jd)                                                          low_rank_calculations.py:51-52
    double p0, p1;
    base_p(p, &p0, &p1);
    const double* spec;
    CHK(spectral_table(c, p, &spec));
    hipLaunchKernelGGL(base_kernel_matrix_kernel<double>, dim3(grid_for(int64_t(cc) * cc)), dim3(256), 0, c->stream, static_cast<const double*>(st->S),
                       static_cast<const double*>(st->S), int64_t(cc), int64_t(cc), d_eff, int(p->base_kernel), p0, p1, spec, st->W);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(add_diag_kernel, dim3((cc + 255) / 256), dim3(256), 0, c->stream, st->W, static_cast<const double*>(st->jd), cc);
    HIPCHK(c, hipGetLastError());
    // eigendecomposition (:55), sign convention, U / sqrt(S + jitter) (:56-57, :60)
    if (cc <= LR_JACOBI_MAX && c->lr_jacobi != 0) {
        const size_t lds = sizeof(double) * (2 * size_t(cc) * (cc + 1) + 2 * size_t(LR_JACOBI_MAX));
        // (the kernel has read W into LDS long before it writes the eigenvectors over it)
        hipLaunchKernelGGL(lr_jacobi_eig_kernel, dim3(1), dim3(LR_JACOBI_THREADS), lds, c->stream, static_cast<const double*>(st->W), cc, st->W, st->ev, st->info);
        HIPCHK(c, hipGetLastError());
    } else {
        std::string err;
        if (!solver_dsyevd(&c->blas_handle, c->stream, cc, st->W, st->ev, st->work, st->info, &err)) return fail(c, GPSIG_ERR_HIP, "%s", err.c_str());
    }
    hipLaunchKernelGGL(eig_sign_kernel, dim3((cc + 63) / 64), dim3(64), 0, c->stream, static_cast<const double*>(st->W), cc, st->work);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(whiten_kernel, dim3(unsigned((int64_t(cc) * cc + 255) / 256)), dim3(256), 0, c->stream, static_cast<const double*>(st->W),
                       static_cast<const double*>(st->ev), static_cast<const double*>(st->work), cc, p->jitter, st->Wh);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(lr_transpose_kernel, dim3(unsigned((cc * cc + 255) / 256)), dim3(256), 0, c->stream, static_cast<const double*>(st->Wh), cc, st->WhT);
    HIPCHK(c, hipGetLastError());
    // one projection per level 2..M (signature_algs.py:184-191)
    for (int i = 0; i < nsk; ++i) {
        gpsig_lr_state::Sk& k = st->sk[i];
        const int64_t D = int64_t(k.k1) * k.k2;
        if (sparsity == 2) {
            hipLaunchKernelGGL(lr_draw_lin_kernel, dim3(1), dim3(64), 0, ss, D, st->r, k.k1, key, uint32_t(i), k.colptr, k.i1, k.i2, k.val, k.ent);
            HIPCHK(c, hipGetLastError());
            continue;
        }
        const double sv = sparsity == 0 ? sqrt(double(D)) : double(D) / log(double(D));       // low_rank_calculations.py:172-175
        const double inv_s = sv < 1.0 ? 1.0 : 1.0 / sv, scale = sqrt((sv < 1.0 ? 1.0 : sv) / double(st->r));   // :192
        for (int fill = 0; fill < 2; ++fill) {
            hipLaunchKernelGGL(lr_draw_sparse_kernel, dim3(unsigned(st->r)), dim3(64), 0, ss, D, st->r, k.k1, inv_s, scale, key, uint32_t(i), fill,
                               k.cap, k.counts, static_cast<const int32_t*>(k.colptr), k.i1, k.i2, k.val, k.ent);
            HIPCHK(c, hipGetLastError());
            if (!fill) {
                hipLaunchKernelGGL(lr_colptr_kernel, dim3(1), dim3(64), 0, ss, static_cast<const int32_t*>(k.counts), st->r, k.cap, k.colptr,
                                   st->info + 1);
                HIPCHK(c, hipGetLastError());
            }
        }
    }
    HIPCHK(c, hipEventRecord(c->side_join, ss));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->side_join, 0));
    // A failed draw (eigensolver not converged, a projection over its capacity) must not give silently wrong features: nothing on the
    // evaluation path reads the flags back (that would be a host synchronisation per evaluation), so the whitening is turned into NaNs
    // and every feature, product and covariance computed from this state is NaN; gpsig_lr_state_sizes reports the cause.
    hipLaunchKernelGGL(lr_poison_kernel, dim3(unsigned((int64_t(cc) * cc + 255) / 256)), dim3(256), 0, c->stream, static_cast<const int*>(st->info), cc, st->Wh, st->WhT);
    HIPCHK(c, hipGetLastError());
    return GPSIG_OK;
}

void gpsig_lr_state_destroy(gpsig_lr_state* st) {
    if (!st) return;
    if (gpsig_ctx* c = st->ctx) {            // still attached: wait for what reads the block, leave the context's registry
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        for (size_t i = 0; i < c->lr_states.size(); ++i)
            if (c->lr_states[i] == st) { c->lr_states.erase(c->lr_states.begin() + i); break; }
    } else {
        (void)hipSetDevice(st->device);      // the context went first (it synchronised its streams then): only the block is left
    }
    if (st->block) (void)hipFree(st->block);
    delete st;
}

// sizes[0..4] = c, d_eff, r, number of projections, Jacobi sweeps taken (0: rocSOLVER); nnz[i] = entries of projection i.  Waits for the draw.
int gpsig_lr_state_sizes(gpsig_ctx* c, const gpsig_lr_state* st, int32_t* sizes, int32_t* nnz) {
    if (!c || !st || !sizes) return GPSIG_ERR_INVALID;
    if (st->ctx != c) return fail(c, GPSIG_ERR_INVALID, "the low-rank state belongs to another (or a destroyed) context");
    HIPCHK(c, hipSetDevice(c->device));
    sizes[0] = st->c; sizes[1] = st->d_eff; sizes[2] = st->r; sizes[3] = st->nsk; sizes[4] = 0;
    int info[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(info, st->info, sizeof(info), hipMemcpyDeviceToHost, c->stream));
    for (int i = 0; i < st->nsk && nnz; ++i)
        HIPCHK(c, hipMemcpyAsync(&nnz[i], st->sk[i].colptr + st->r, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    sizes[4] = info[2];
    if (info[0] != 0) return fail(c, GPSIG_ERR_HIP, "the eigendecomposition of the landmark Gram did not converge (info = %d)", info[0]);
    if (info[1] != 0) return fail(c, GPSIG_ERR_HIP, "a random projection drew more entries than its capacity (a 12-sigma event: draw again)");
    return GPSIG_OK;
}

// Copies of what was drawn, all HOST pointers: landmarks (c, d'), jitter_diag (c), whitening (c, c), eigenvalues (c) -- any may be
// NULL --, and for projection i the arrays of sketches[i] (colptr r+1, i1 / i2 / val nnz[i] as reported by gpsig_lr_state_sizes).
int gpsig_lr_state_export(gpsig_ctx* c, const gpsig_lr_state* st, double* landmarks, double* jitter_diag, double* whitening, double* eigenvalues,
                          const gpsig_sketch* sketches) {
    if (!c || !st) return GPSIG_ERR_INVALID;
    if (st->ctx != c) return fail(c, GPSIG_ERR_INVALID, "the low-rank state belongs to another (or a destroyed) context");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t cc = size_t(st->c);
    if (landmarks) HIPCHK(c, hipMemcpyAsync(landmarks, st->S, sizeof(double) * cc * st->d_eff, hipMemcpyDeviceToHost, c->stream));
    if (jitter_diag) HIPCHK(c, hipMemcpyAsync(jitter_diag, st->jd, sizeof(double) * cc, hipMemcpyDeviceToHost, c->stream));
    if (whitening) HIPCHK(c, hipMemcpyAsync(whitening, st->Wh, sizeof(double) * cc * cc, hipMemcpyDeviceToHost, c->stream));
    if (eigenvalues) HIPCHK(c, hipMemcpyAsync(eigenvalues, st->ev, sizeof(double) * cc, hipMemcpyDeviceToHost, c->stream));
    for (int i = 0; i < st->nsk && sketches; ++i) {
        const gpsig_sketch& h = sketches[i];
        const gpsig_lr_state::Sk& k = st->sk[i];
        if (h.nnz < 0 || h.nnz > k.cap) return fail(c, GPSIG_ERR_INVALID, "projection %d: %d entries asked for, capacity %d", i, h.nnz, k.cap);
        HIPCHK(c, hipMemcpyAsync(const_cast<int32_t*>(h.colptr), k.colptr, sizeof(int32_t) * (size_t(st->r) + 1), hipMemcpyDeviceToHost, c->stream));
        if (h.nnz) {
            HIPCHK(c, hipMemcpyAsync(const_cast<int32_t*>(h.i1), k.i1, sizeof(int32_t) * size_t(h.nnz), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(const_cast<int32_t*>(h.i2), k.i2, sizeof(int32_t) * size_t(h.nnz), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(const_cast<double*>(h.val), k.val, sizeof(double) * size_t(h.nnz), hipMemcpyDeviceToHost, c->stream));
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GPSIG_OK;
}


}  // extern "C"
