// grad_fused_inst.hpp -- instantiation list of seq_grad_fused_kernel (grad_fused_kernel.hpp): RBF and the Matern families, padded feature counts
// 4 / 8 (four columns per lane) and 16 (two; with differences only), num_levels 2 .. 6 at compile time.  Included by grad_fused_inst_*.hip, one unit
// per (lanes per pair, difference) so that the build runs them side by side.
#pragma once
#include "grad_fused_kernel.hpp"

namespace gpsig {

typedef hipError_t (*FusedGradLaunchFn)(const FusedGradArgs&, int, size_t, hipStream_t);

template <int DP, int LQ, int KIND, int G, bool DIFF>
static hipError_t fused_grad_launch(const FusedGradArgs& a, int ntasks, size_t lds, hipStream_t s) {
    auto kern = seq_grad_fused_kernel<DP, LQ, KIND, G, fused_grad_columns(DP), DIFF>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(ntasks), dim3(128), lds, s, a);
    return hipGetLastError();
}

template <int KIND, int G, bool DIFF>
static FusedGradLaunchFn fused_grad_lookup_kind(int DP, int LQ) {
#define FG_PICK(D_)                                          \
    if (DP == D_) switch (LQ) {                              \
        case 1: return fused_grad_launch<D_, 1, KIND, G, DIFF>;       \
        case 2: return fused_grad_launch<D_, 2, KIND, G, DIFF>;       \
        case 3: return fused_grad_launch<D_, 3, KIND, G, DIFF>;       \
        case 4: return fused_grad_launch<D_, 4, KIND, G, DIFF>;       \
        case 5: return fused_grad_launch<D_, 5, KIND, G, DIFF>;       \
        default: return nullptr;                             \
    }
    FG_PICK(4)
    FG_PICK(8)
    if constexpr (DIFF) { FG_PICK(16) }          // (difference=False is built for up to 8 columns of state space)
#undef FG_PICK
    return nullptr;
}

template <int G, bool DIFF>
static FusedGradLaunchFn fused_grad_lookup_g(int kind, int DP, int LQ) {
    switch (kind) {
        case BASE_RBF: return fused_grad_lookup_kind<BASE_RBF, G, DIFF>(DP, LQ);
        case BASE_MATERN12: return fused_grad_lookup_kind<BASE_MATERN12, G, DIFF>(DP, LQ);
        case BASE_MATERN32: return fused_grad_lookup_kind<BASE_MATERN32, G, DIFF>(DP, LQ);
        case BASE_MATERN52: return fused_grad_lookup_kind<BASE_MATERN52, G, DIFF>(DP, LQ);
        default: return nullptr;
    }
}

}  // namespace gpsig
