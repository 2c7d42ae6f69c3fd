// reverse pass of the tensor-vs-sequence chains, tile kernel: the HIGHER-ORDER chains (signature_algs.py:129-160) of SignatureRBF at a run-time order,
// num_levels 3, 4, 5, feature widths 4, 6, 8 (the forward instances of tvs_tile_inst_ho.hip leave the chain totals these continue from)
#include "tvs_grad_tile_kernel.hpp"

namespace gpsig {
typedef hipError_t (*TvsGradTileLaunchFn)(const TvsGradTileArgs&, dim3, size_t, hipStream_t);

template <int M, int D, bool PAIRED, int KIND = BASE_RBF>
static hipError_t tvs_grad_tile_launch_ho(const TvsGradTileArgs& A, dim3 grid, size_t lds, hipStream_t stream) {
    auto kern = tvs_grad_tile_kernel<M, D, KIND, PAIRED, true>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64), lds, stream, A);
    return hipGetLastError();
}

template <int M, int KIND>
static TvsGradTileLaunchFn pick_ho(int D, bool paired) {
    if (D == 4) return paired ? &tvs_grad_tile_launch_ho<M, 4, true, KIND> : &tvs_grad_tile_launch_ho<M, 4, false, KIND>;
    if (D == 6) return paired ? &tvs_grad_tile_launch_ho<M, 6, true, KIND> : &tvs_grad_tile_launch_ho<M, 6, false, KIND>;
    if (D == 8) return paired ? &tvs_grad_tile_launch_ho<M, 8, true, KIND> : &tvs_grad_tile_launch_ho<M, 8, false, KIND>;
    return nullptr;
}

// kind: BASE_RBF or TVSG_MATERN (the three Matern families as one instruction stream)
TvsGradTileLaunchFn tvs_grad_tile_lookup_ho(int M, int D, bool paired, int kind) {
    if (kind == TVSG_MATERN) {
        switch (M) {
            case 3: return pick_ho<3, TVSG_MATERN>(D, paired);
            case 4: return pick_ho<4, TVSG_MATERN>(D, paired);
            case 5: return pick_ho<5, TVSG_MATERN>(D, paired);
            default: return nullptr;
        }
    }
    switch (M) {
        case 3: return pick_ho<3, BASE_RBF>(D, paired);
        case 4: return pick_ho<4, BASE_RBF>(D, paired);
        case 5: return pick_ho<5, BASE_RBF>(D, paired);
        default: return nullptr;
    }
}
}  // namespace gpsig
