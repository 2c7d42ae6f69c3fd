// lr_fused_args.hpp -- argument block and launcher of the fused low-rank feature kernel (lr_fused_kernel.hpp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aux_kernels.hpp"

namespace gpsig {

// Landmarks, whitening matrix and sketch entries are read-only for the whole launch and addressed wave-uniformly: in the
// constant address space the compiler may serve them through the scalar unit (s_load) instead of broadcasting vector loads.
template <typename T>
using lr_const_ptr = const __attribute__((address_space(4))) T*;
template <typename T>
__device__ __forceinline__ lr_const_ptr<T> lr_as_const(const T* p) { return (lr_const_ptr<T>)(p); }

struct LrEntry { double val; int32_t i1, i2; };     // one entry of a sketch, stored by output column (16 bytes: one s_load_dwordx4)

struct LrFusedSketch { const int32_t* colptr; const LrEntry* ent; };

constexpr int LR_FUSED_MAX_SKETCHES = 7;            // levels 2 .. 8
constexpr size_t LR_FUSED_MAX_LDS = 156 * 1024;     // of the 160 KB a CU has (one workgroup per CU at that size)

struct LrFusedArgs {
    const double* X; int64_t N; int L;
    ScaleParams P;
    const double* S;        // landmarks (c, d_eff), scaled points
    const double* Wh;       // whitening (c, c) row-major: feat[j] = sum_i kxs[i] * Wh[i][j]
    int c, r, M, difference, kind;
    double p0, p1;
    LrFusedSketch sk[LR_FUSED_MAX_SKETCHES];
    double* Phi; int F;
    int lp;                 // row stride of the LDS arrays, in doubles (odd, > number of time steps rounded up to 64)
    int rows_b;             // rows of the two work arrays: max(c, r, d_eff)
};

inline int lr_fused_stride(int L, int pad) { return (L + 63) / 64 * 64 + pad; }
inline size_t lr_fused_lds_bytes(int c, int r, int d_eff, int L, int pad = 1) {
    const int lp = lr_fused_stride(L, pad);
    int kb = c > r ? c : r;
    if (d_eff > kb) kb = d_eff;
    return sizeof(double) * size_t(lp) * (size_t(c) + 2 * size_t(kb));
}

// Feature map of inducing tensors (gpsig/kernels.py:285-311 _K_tens_lr_feat, signature_algs.py:194-222 tensor_kern_lr_feature),
// one workgroup per tensor: its lt * E components are whitened and chained through the sketches in LDS.
struct LrTensFusedArgs {
    const double* Z; int64_t T; int lt, E;      // Z (lt, T, E, d_eff) as the caller gives it
    ScaleParams P;
    const double* S; const double* Wh;
    int c, r, M, kind;
    double p0, p1;
    LrFusedSketch sk[LR_FUSED_MAX_SKETCHES];
    double* Phi; int F;
};
inline size_t lr_tens_fused_lds_bytes(int c, int r, int d_eff, int lt, int E) {
    const size_t rows = size_t(lt) * E, w = size_t(c > r ? c : r);
    return sizeof(double) * (rows * (size_t(d_eff) + 2 * size_t(c)) + size_t(lt) * c + 2 * w);
}
int lr_tens_fused_launch(hipStream_t stream, const LrTensFusedArgs& A);

// two-array form (lr_seq_features_fused2_kernel): usable for L <= 64 and at most 8 output columns per wavefront of the 512-thread workgroup
inline bool lr_fused2_ok(int c, int r, int L) { return L <= 64 && c <= 64 && r <= 64; }
inline size_t lr_fused2_lds_bytes(int c, int r, int d_eff, int L, int pad = 1) {
    int kb = c > r ? c : r;
    if (d_eff > kb) kb = d_eff;
    return sizeof(double) * size_t(lr_fused_stride(L, pad)) * 2 * size_t(kb);
}
int lr_fused2_launch(hipStream_t stream, const LrFusedArgs& A, unsigned grid);

// lr_fused_inst.hip: launches the kernel on `stream` with `grid` workgroups; returns the hipError_t of the launch
int lr_fused_launch(hipStream_t stream, const LrFusedArgs& A, unsigned grid, int variant);

}  // namespace gpsig
