// seq_grad_fused_kernel instances (grad_fused_kernel.hpp): RBF and the Matern families, padded feature counts 4 / 8, num_levels 2 .. 6 at compile time
#include "grad_fused_kernel.hpp"

namespace gpsig {

typedef hipError_t (*FusedGradLaunchFn)(const FusedGradArgs&, int, size_t, hipStream_t);

template <int DP, int LQ, int KIND>
static hipError_t fused_grad_launch(const FusedGradArgs& a, int ntasks, size_t lds, hipStream_t s) {
    auto kern = seq_grad_fused_kernel<DP, LQ, KIND>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(ntasks), dim3(128), lds, s, a);
    return hipGetLastError();
}

template <int KIND>
static FusedGradLaunchFn fused_grad_lookup_kind(int DP, int LQ) {
#define FG_PICK(D_)                                          \
    if (DP == D_) switch (LQ) {                              \
        case 1: return fused_grad_launch<D_, 1, KIND>;       \
        case 2: return fused_grad_launch<D_, 2, KIND>;       \
        case 3: return fused_grad_launch<D_, 3, KIND>;       \
        case 4: return fused_grad_launch<D_, 4, KIND>;       \
        case 5: return fused_grad_launch<D_, 5, KIND>;       \
        default: return nullptr;                             \
    }
    FG_PICK(4)
    FG_PICK(8)
#undef FG_PICK
    return nullptr;
}

FusedGradLaunchFn fused_grad_lookup(int kind, int DP, int LQ) {
    switch (kind) {
        case BASE_RBF: return fused_grad_lookup_kind<BASE_RBF>(DP, LQ);
        case BASE_MATERN12: return fused_grad_lookup_kind<BASE_MATERN12>(DP, LQ);
        case BASE_MATERN32: return fused_grad_lookup_kind<BASE_MATERN32>(DP, LQ);
        case BASE_MATERN52: return fused_grad_lookup_kind<BASE_MATERN52>(DP, LQ);
        default: return nullptr;
    }
}

}  // namespace gpsig
