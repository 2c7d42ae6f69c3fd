// sig_feat_pick.hpp -- launchers of the feature kernels (sig_feat_kernel.hpp) for one range of column counts; included by the
// sig_feat_inst_*.hip translation units, which are compiled in parallel.
#pragma once

#include "sig_feat_kernel.hpp"

namespace gpsig {
typedef hipError_t (*SigFeatLaunchFn)(const SigFeatArgs&, unsigned, size_t, hipStream_t);

template <int D, int M>
static hipError_t sig_feat_launch(const SigFeatArgs& A, unsigned grid, size_t lds, hipStream_t stream) {
    auto kern = sig_features_kernel<D, M, false>;
    // sibling parents per thread: fewer multiply-adds, levels stored in a line-friendly order of their own (natural_order: the reverse
    // pass of sig_feat_grad_kernel.hpp reads the features by index)
    if constexpr (sig_siblings(D, M)) if (!A.natural_order) kern = sig_features_sib_kernel<D, M, false>;
    if (A.order > 1) {                               // the higher-order algorithm: truncated-exponential steps
        kern = sig_features_kernel<D, M, true>;
        if constexpr (sig_siblings(D, M)) if (!A.natural_order) kern = sig_features_sib_kernel<D, M, true>;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(sig_threads(D, M)), lds, stream, A);
    return hipGetLastError();
}

template <int D>
static SigFeatLaunchFn sig_feat_pick(int M) {
    switch (M) {
        case 2: return &sig_feat_launch<D, 2>;
        case 3: return &sig_feat_launch<D, 3>;
        case 4: if constexpr (sig_ipow(D, 4) <= SIG_MAX_TOP) return &sig_feat_launch<D, 4>; else return nullptr;
        case 5: if constexpr (sig_ipow(D, 5) <= SIG_MAX_TOP) return &sig_feat_launch<D, 5>; else return nullptr;
        case 6: if constexpr (sig_ipow(D, 5) <= SIG_MAX_TOP && sig_ipow(D, 6) <= SIG_MAX_TOP) return &sig_feat_launch<D, 6>; else return nullptr;
        case 7: if constexpr (sig_ipow(D, 6) <= SIG_MAX_TOP && sig_ipow(D, 7) <= SIG_MAX_TOP) return &sig_feat_launch<D, 7>; else return nullptr;
        case 8: if constexpr (sig_ipow(D, 7) <= SIG_MAX_TOP && sig_ipow(D, 8) <= SIG_MAX_TOP) return &sig_feat_launch<D, 8>; else return nullptr;
        default: return nullptr;
    }
}

}  // namespace gpsig
