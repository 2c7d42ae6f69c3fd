// Higher-order instances of the Kzx tile kernel (round 6): the chains of signature_algs.py:129-160 for the RBF kernel, num_levels 3 / 4 / 5, feature widths
// 4 / 6 / 8, with and without increments; the order is a run-time argument (>= 2).  Level sets as the first-order instances' (the state between time steps is
// the same: the running totals).
#include "tvs_tile_kernel.hpp"

namespace gpsig {
typedef hipError_t (*TvsTileLaunchFn)(TvsTileArgs&, size_t, hipStream_t, int);

template <int M, int NW, int D, bool INCR>
static hipError_t tvs_tile_launch_ho(TvsTileArgs& A, size_t lds, hipStream_t stream, int num_cus) {
    auto kern = tvs_tile_kernel<M, NW, D, INCR, BASE_RBF, true>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, TVS_WG_WAVES * 64, lds);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    const int64_t TB = A.Tpad / 64, slots = int64_t(per_cu) * (num_cus > 0 ? num_cus : 256), workers = slots * TVS_WG_WAVES;
    A.plan_items(workers / TB > 0 ? workers / TB : 1);
    const int64_t all_items = TB * int64_t(A.items), need = (all_items + TVS_WG_WAVES - 1) / TVS_WG_WAVES;
    dim3 grid((unsigned)(slots < need ? slots : need));
    hipLaunchKernelGGL(kern, grid, dim3(TVS_WG_WAVES * 64), lds, stream, A);
    return hipGetLastError();
}

template <int M, int D, bool INCR>
static TvsTileLaunchFn ho_pick(int NW) {
    constexpr int P = tvs_planned_sets(M, D, INCR, BASE_RBF);
    if constexpr (P > 0) {
        if (NW == P) return &tvs_tile_launch_ho<M, P, D, INCR>;
    }
    return nullptr;
}

// RBF, the planner's number of level sets (tvs_planned_sets) only
TvsTileLaunchFn tvs_tile_lookup_ho(int M, int NW, int D, bool incr) {
#define TVS_HO_CASE(M_, D_) if (M == M_ && D == D_) return incr ? ho_pick<M_, D_, true>(NW) : ho_pick<M_, D_, false>(NW);
    TVS_HO_CASE(3, 4) TVS_HO_CASE(3, 6) TVS_HO_CASE(3, 8) TVS_HO_CASE(4, 4) TVS_HO_CASE(4, 6) TVS_HO_CASE(4, 8) TVS_HO_CASE(5, 4) TVS_HO_CASE(5, 6) TVS_HO_CASE(5, 8)
#undef TVS_HO_CASE
    return nullptr;
}
}  // namespace gpsig
