// seq_grad_fused_kernel instances that run the backward sweep only, from the stash the evaluation kernel wrote (seq_inst_ptdrbf_stash.hip: RBF and the Matern
// families with differences, 16 lanes per pair, four columns per lane, 4 / 8 padded features, num_levels 4 / 5)
#include "grad_fused_kernel.hpp"

namespace gpsig {
typedef hipError_t (*FusedGradLaunchFn)(const FusedGradArgs&, int, size_t, hipStream_t);

template <int DP, int LQ, int KIND>
static hipError_t fused_grad_stash_launch(const FusedGradArgs& a, int ntasks, size_t lds, hipStream_t s) {
    auto kern = seq_grad_fused_kernel<DP, LQ, KIND, 16, 4, true, 2>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(ntasks), dim3(128), lds, s, a);
    return hipGetLastError();
}

template <int KIND>
static FusedGradLaunchFn fused_grad_stash_lookup_kind(int DP, int LQ) {
    if (DP == 4 && LQ == 3) return fused_grad_stash_launch<4, 3, KIND>;
    if (DP == 4 && LQ == 4) return fused_grad_stash_launch<4, 4, KIND>;
    if (DP == 8 && LQ == 3) return fused_grad_stash_launch<8, 3, KIND>;
    if (DP == 8 && LQ == 4) return fused_grad_stash_launch<8, 4, KIND>;
    return nullptr;
}

FusedGradLaunchFn fused_grad_stash_lookup(int kind, int DP, int LQ) {
    switch (kind) {
        case BASE_RBF: return fused_grad_stash_lookup_kind<BASE_RBF>(DP, LQ);
        case BASE_MATERN12: return fused_grad_stash_lookup_kind<BASE_MATERN12>(DP, LQ);
        case BASE_MATERN32: return fused_grad_stash_lookup_kind<BASE_MATERN32>(DP, LQ);
        case BASE_MATERN52: return fused_grad_stash_lookup_kind<BASE_MATERN52>(DP, LQ);
        default: return nullptr;
    }
}
}  // namespace gpsig
