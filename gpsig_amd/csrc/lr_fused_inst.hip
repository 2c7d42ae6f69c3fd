// lr_fused_inst.hip -- the fused low-rank feature kernel and its launcher (own translation unit: api.hip only sees the arguments).
#include "lr_fused_kernel.hpp"

namespace gpsig {

int lr_fused_launch(hipStream_t stream, const LrFusedArgs& A, unsigned grid) {
    const size_t lds = lr_fused_lds_bytes(A.c, A.r, A.P.d_eff(), A.L);
    static size_t allowed = 0;                       // dynamic LDS beyond 64 KB has to be requested once per process
    if (lds > allowed) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lr_seq_features_fused_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return int(e);
        allowed = lds;
    }
    hipLaunchKernelGGL(lr_seq_features_fused_kernel, dim3(grid), dim3(LR_FUSED_THREADS), lds, stream, A);
    return int(hipGetLastError());
}

}  // namespace gpsig
