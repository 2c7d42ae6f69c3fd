// One translation unit of tvs_tile_kernel instantiations: #define TVS_TILE_M and TVS_TILE_NWS(X) (the waves-per-workgroup
// values built for that num_levels) before including.  Feature widths: 4, 6, 8.
#include "tvs_tile_kernel.hpp"

namespace gpsig {
typedef hipError_t (*TvsTileLaunchFn)(TvsTileArgs&, size_t, hipStream_t, int);

// Persistent launch: as many workgroups as the chip holds at once (the occupancy the runtime reports for this instance and its LDS), each of
// their four wavefronts drawing (tensor block, run of sequences) items from the per-block counters A.queue -- long runs first, short ones last
// (TvsTileArgs::plan_items).  NW: the number of level sets a wavefront sweeps a tile of sequences in.
template <int M, int NW, int D, bool INCR, int KIND>
static hipError_t tvs_tile_launch(TvsTileArgs& A, size_t lds, hipStream_t stream, int num_cus) {
    auto kern = tvs_tile_kernel<M, NW, D, INCR, KIND>;
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

// The Matern families are built for the number of level sets the planner takes only (tvs_planned_sets; the option tvs_tile_nw does not apply to them).
template <int M, int NW, int D, bool INCR, int KIND>
static TvsTileLaunchFn tvs_tile_planned() {
    if constexpr (NW == tvs_planned_sets(M, D, INCR, KIND)) return &tvs_tile_launch<M, NW, D, INCR, KIND>;
    else return nullptr;
}
template <int M, int NW, int D>
static TvsTileLaunchFn tvs_tile_pick(bool incr, int kind) {
    if (kind == BASE_LINEAR) return &tvs_tile_launch<M, NW, D, false, BASE_LINEAR>;       // increments arrive collapsed
    if (kind == BASE_RBF) return incr ? &tvs_tile_launch<M, NW, D, true, BASE_RBF> : &tvs_tile_launch<M, NW, D, false, BASE_RBF>;
    if (kind == BASE_MATERN12) return incr ? tvs_tile_planned<M, NW, D, true, BASE_MATERN12>() : tvs_tile_planned<M, NW, D, false, BASE_MATERN12>();
    if (kind == BASE_MATERN32) return incr ? tvs_tile_planned<M, NW, D, true, BASE_MATERN32>() : tvs_tile_planned<M, NW, D, false, BASE_MATERN32>();
    if (kind == BASE_MATERN52) return incr ? tvs_tile_planned<M, NW, D, true, BASE_MATERN52>() : tvs_tile_planned<M, NW, D, false, BASE_MATERN52>();
    return incr ? &tvs_tile_launch<M, NW, D, true, -1> : &tvs_tile_launch<M, NW, D, false, -1>;
}

#define TVS_TILE_CAT2(a, b) a##b
#define TVS_TILE_CAT(a, b) TVS_TILE_CAT2(a, b)
// kind: BASE_LINEAR, BASE_RBF, a Matern family or -1 (any other family, evaluated by base_eval_n at run time)
TvsTileLaunchFn TVS_TILE_CAT(tvs_tile_lookup_m, TVS_TILE_M)(int NW, int D, bool incr, int kind) {
#define TVS_TILE_CASE(NW_)                                                              \
    if (NW == NW_) {                                                                    \
        if (D == 4) return tvs_tile_pick<TVS_TILE_M, NW_, 4>(incr, kind);               \
        if (D == 6) return tvs_tile_pick<TVS_TILE_M, NW_, 6>(incr, kind);               \
        if (D == 8) return tvs_tile_pick<TVS_TILE_M, NW_, 8>(incr, kind);               \
    }
    TVS_TILE_NWS(TVS_TILE_CASE)
#undef TVS_TILE_CASE
    return nullptr;
}
}  // namespace gpsig
