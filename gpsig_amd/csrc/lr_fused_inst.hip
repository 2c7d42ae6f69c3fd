// lr_fused_inst.hip -- the fused low-rank feature kernel and its launcher (own translation unit: api.hip only sees the arguments).
#include "lr_fused_kernel.hpp"

namespace gpsig {

namespace {
template <int THREADS, int UNROLL>
int launch(hipStream_t stream, const LrFusedArgs& A, unsigned grid, size_t lds) {
    if (lds > 48 * 1024) {                           // dynamic LDS beyond the default has to be requested: per launch (no process-wide cache of what
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lr_seq_features_fused_kernel<THREADS, UNROLL>),   // one device was granted)
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return int(e);
    }
    hipLaunchKernelGGL((lr_seq_features_fused_kernel<THREADS, UNROLL>), dim3(grid), dim3(THREADS), lds, stream, A);
    return int(hipGetLastError());
}
}  // namespace

int lr_fused_launch(hipStream_t stream, const LrFusedArgs& A, unsigned grid, int variant) {
    const size_t lds = sizeof(double) * size_t(A.lp) * (size_t(A.c) + 2 * size_t(A.rows_b));
    // BASELINE configs[2]'s sequences (L=50, c=r=50, 'sqrt'), same box: 256 threads / 4 entries per batch 4.32 ms, 256 / 8 3.84,
    // 512 / 4 2.71, 512 / 8 2.57, 1024 / 4 3.09, 1024 / 8 4.01 (profiles/r02_lowrank.txt)
    switch (variant) {
        case 1: return launch<256, 4>(stream, A, grid, lds);
        case 2: return launch<256, 8>(stream, A, grid, lds);
        case 3: return launch<512, 4>(stream, A, grid, lds);
        default: return launch<512, 8>(stream, A, grid, lds);
    }
}

int lr_fused2_launch(hipStream_t stream, const LrFusedArgs& A, unsigned grid) {
    const size_t lds = sizeof(double) * size_t(A.lp) * 2 * size_t(A.rows_b);
    auto kern = lr_seq_features_fused2_kernel<512, 8>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return int(e);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, A);
    return int(hipGetLastError());
}

int lr_tens_fused_launch(hipStream_t stream, const LrTensFusedArgs& A) {
    const size_t lds = lr_tens_fused_lds_bytes(A.c, A.r, A.P.d_eff(), A.lt, A.E);
    hipLaunchKernelGGL(lr_tens_features_fused_kernel, dim3(unsigned(A.T)), dim3(LR_TENS_THREADS), lds, stream, A);
    return int(hipGetLastError());
}

}  // namespace gpsig
