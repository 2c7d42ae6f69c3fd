// sig_feat_grad_pick.hpp -- launchers of the features' reverse pass (sig_feat_grad_kernel.hpp) for one range of column counts; included by
// the sig_feat_grad_inst_*.hip translation units, which are compiled in parallel (the same shapes as sig_feat_pick.hpp).
#pragma once

#include "sig_feat_grad_kernel.hpp"

namespace gpsig {
typedef hipError_t (*SigFeatGradLaunchFn)(const SigFeatGradArgs&, unsigned, size_t, hipStream_t);

template <int D, int M>
static hipError_t sig_feat_grad_launch(const SigFeatGradArgs& A, unsigned grid, size_t lds, hipStream_t stream) {
    auto kern = sig_feat_reverse_kernel<D, M>;
    if (A.order > 1) kern = sig_feat_reverse_ho_kernel<D, M>;       // the higher-order algorithm: Horner sub-steps (one more LDS layout: the caller sized `lds` for it)
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(sig_threads(D, M)), lds, stream, A);
    return hipGetLastError();
}

template <int D>
static SigFeatGradLaunchFn sig_feat_grad_pick(int M) {
    switch (M) {
        case 2: return &sig_feat_grad_launch<D, 2>;
        case 3: return &sig_feat_grad_launch<D, 3>;
        case 4: if constexpr (sig_ipow(D, 4) <= SIG_MAX_TOP) return &sig_feat_grad_launch<D, 4>; else return nullptr;
        case 5: if constexpr (sig_ipow(D, 5) <= SIG_MAX_TOP) return &sig_feat_grad_launch<D, 5>; else return nullptr;
        case 6: if constexpr (sig_ipow(D, 5) <= SIG_MAX_TOP && sig_ipow(D, 6) <= SIG_MAX_TOP) return &sig_feat_grad_launch<D, 6>; else return nullptr;
        case 7: if constexpr (sig_ipow(D, 6) <= SIG_MAX_TOP && sig_ipow(D, 7) <= SIG_MAX_TOP) return &sig_feat_grad_launch<D, 7>; else return nullptr;
        case 8: if constexpr (sig_ipow(D, 7) <= SIG_MAX_TOP && sig_ipow(D, 8) <= SIG_MAX_TOP) return &sig_feat_grad_launch<D, 8>; else return nullptr;
        default: return nullptr;
    }
}

}  // namespace gpsig
