// lr_fused_args.hpp -- argument block and launcher of the fused low-rank feature kernel (lr_fused_kernel.hpp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aux_kernels.hpp"

namespace gpsig {

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

inline size_t lr_fused_lds_bytes(int c, int r, int d_eff, int L) {
    const int lp = (L + 63) / 64 * 64 + 1;
    int kb = c > r ? c : r;
    if (d_eff > kb) kb = d_eff;
    return sizeof(double) * size_t(lp) * (size_t(c) + 2 * size_t(kb));
}

// lr_fused_inst.hip: launches the kernel on `stream` with `grid` workgroups; returns the hipError_t of the launch
int lr_fused_launch(hipStream_t stream, const LrFusedArgs& A, unsigned grid, int variant);

}  // namespace gpsig
