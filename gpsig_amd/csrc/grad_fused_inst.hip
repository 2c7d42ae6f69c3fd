// seq_grad_fused_kernel instances (grad_fused_kernel.hpp): padded feature counts 4 / 8, num_levels 2 .. 6 at compile time
#include "grad_fused_kernel.hpp"

namespace gpsig {

typedef hipError_t (*FusedGradLaunchFn)(const FusedGradArgs&, int, size_t, hipStream_t);

template <int DP, int LQ>
static hipError_t fused_grad_launch(const FusedGradArgs& a, int ntasks, size_t lds, hipStream_t s) {
    auto kern = seq_grad_fused_kernel<DP, LQ>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(ntasks), dim3(128), lds, s, a);
    return hipGetLastError();
}

FusedGradLaunchFn fused_grad_lookup(int DP, int LQ) {
#define FG_PICK(D_)                                    \
    if (DP == D_) switch (LQ) {                        \
        case 1: return fused_grad_launch<D_, 1>;       \
        case 2: return fused_grad_launch<D_, 2>;       \
        case 3: return fused_grad_launch<D_, 3>;       \
        case 4: return fused_grad_launch<D_, 4>;       \
        case 5: return fused_grad_launch<D_, 5>;       \
        default: return nullptr;                       \
    }
    FG_PICK(4)
    FG_PICK(8)
#undef FG_PICK
    return nullptr;
}

}  // namespace gpsig
