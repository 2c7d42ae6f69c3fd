// One translation unit of tvs_grad_tile_kernel instantiations: #define TVSG_M before including.  Feature widths 4, 6, 8; the
// base kernel linear / RBF / run-time family; incremental tensors of the non-linear families as lane pairs.
#include "tvs_grad_tile_kernel.hpp"

namespace gpsig {
typedef hipError_t (*TvsGradTileLaunchFn)(const TvsGradTileArgs&, dim3, size_t, hipStream_t);

template <int M, int D, int KIND, bool PAIRED>
static hipError_t tvs_grad_tile_launch(const TvsGradTileArgs& A, dim3 grid, size_t lds, hipStream_t stream) {
    auto kern = tvs_grad_tile_kernel<M, D, KIND, PAIRED>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64), lds, stream, A);
    return hipGetLastError();
}

template <int M, int D>
static TvsGradTileLaunchFn tvs_grad_tile_pick(int kind, bool paired) {
    if (kind == BASE_LINEAR) return paired ? nullptr : &tvs_grad_tile_launch<M, D, BASE_LINEAR, false>;     // increments arrive collapsed
    if (kind == BASE_RBF) return paired ? &tvs_grad_tile_launch<M, D, BASE_RBF, true> : &tvs_grad_tile_launch<M, D, BASE_RBF, false>;
    if (kind == TVSG_MATERN) return paired ? &tvs_grad_tile_launch<M, D, TVSG_MATERN, true> : &tvs_grad_tile_launch<M, D, TVSG_MATERN, false>;
    return paired ? &tvs_grad_tile_launch<M, D, -1, true> : &tvs_grad_tile_launch<M, D, -1, false>;
}

#define TVSG_CAT2(a, b) a##b
#define TVSG_CAT(a, b) TVSG_CAT2(a, b)
TvsGradTileLaunchFn TVSG_CAT(tvs_grad_tile_lookup_m, TVSG_M)(int D, int kind, bool paired) {
#ifndef TVSG_ONLY_D6
    if (D == 4) return tvs_grad_tile_pick<TVSG_M, 4>(kind, paired);
    if (D == 8) return tvs_grad_tile_pick<TVSG_M, 8>(kind, paired);
#endif
    if (D == 6) return tvs_grad_tile_pick<TVSG_M, 6>(kind, paired);
    return nullptr;
}
}  // namespace gpsig
