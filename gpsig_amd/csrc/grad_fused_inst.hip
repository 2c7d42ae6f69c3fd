// seq_grad_fused_kernel instances (grad_fused_kernel.hpp): RBF and the Matern families, 16 or 64 lanes per pair, padded feature counts 4 / 8 (four columns per lane) and 16 (two), num_levels 2 .. 6 at compile time
#include "grad_fused_kernel.hpp"

namespace gpsig {

typedef hipError_t (*FusedGradLaunchFn)(const FusedGradArgs&, int, size_t, hipStream_t);

template <int DP, int LQ, int KIND, int G>
static hipError_t fused_grad_launch(const FusedGradArgs& a, int ntasks, size_t lds, hipStream_t s) {
    auto kern = seq_grad_fused_kernel<DP, LQ, KIND, G, fused_grad_columns(DP)>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(ntasks), dim3(128), lds, s, a);
    return hipGetLastError();
}

template <int KIND, int G>
static FusedGradLaunchFn fused_grad_lookup_kind(int DP, int LQ) {
#define FG_PICK(D_)                                          \
    if (DP == D_) switch (LQ) {                              \
        case 1: return fused_grad_launch<D_, 1, KIND, G>;       \
        case 2: return fused_grad_launch<D_, 2, KIND, G>;       \
        case 3: return fused_grad_launch<D_, 3, KIND, G>;       \
        case 4: return fused_grad_launch<D_, 4, KIND, G>;       \
        case 5: return fused_grad_launch<D_, 5, KIND, G>;       \
        default: return nullptr;                             \
    }
    FG_PICK(4)
    FG_PICK(8)
    FG_PICK(16)
#undef FG_PICK
    return nullptr;
}

template <int G>
static FusedGradLaunchFn fused_grad_lookup_g(int kind, int DP, int LQ) {
    switch (kind) {
        case BASE_RBF: return fused_grad_lookup_kind<BASE_RBF, G>(DP, LQ);
        case BASE_MATERN12: return fused_grad_lookup_kind<BASE_MATERN12, G>(DP, LQ);
        case BASE_MATERN32: return fused_grad_lookup_kind<BASE_MATERN32, G>(DP, LQ);
        case BASE_MATERN52: return fused_grad_lookup_kind<BASE_MATERN52, G>(DP, LQ);
        default: return nullptr;
    }
}

// G: lanes per pair group -- 16 (column side of at most 64 points, four pairs per wavefront) or 64 (at most 256 points, one pair)
FusedGradLaunchFn fused_grad_lookup(int kind, int DP, int LQ, int G) {
    return G == 16 ? fused_grad_lookup_g<16>(kind, DP, LQ) : (G == 64 ? fused_grad_lookup_g<64>(kind, DP, LQ) : nullptr);
}

}  // namespace gpsig
