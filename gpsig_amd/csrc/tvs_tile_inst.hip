// Dispatch over the tvs_tile_kernel translation units (tvs_tile_inst_m*.hip) and the planner's shape rules.
#include "tvs_tile_kernel.hpp"

namespace gpsig {
typedef hipError_t (*TvsTileLaunchFn)(TvsTileArgs&, size_t, hipStream_t, int);
TvsTileLaunchFn tvs_tile_lookup_m2(int, int, bool, int);
TvsTileLaunchFn tvs_tile_lookup_m3(int, int, bool, int);
TvsTileLaunchFn tvs_tile_lookup_m4(int, int, bool, int);
TvsTileLaunchFn tvs_tile_lookup_m5(int, int, bool, int);
TvsTileLaunchFn tvs_tile_lookup_m6(int, int, bool, int);

// feature width the kernel is built for (0: none)
int tvs_tile_width(int d) { return d <= 4 ? 4 : (d <= 6 ? 6 : (d <= 8 ? 8 : 0)); }

TvsTileLaunchFn tvs_tile_lookup(int M, int NW, int D, bool incr, int kind) {
    switch (M) {
        case 2: return tvs_tile_lookup_m2(NW, D, incr, kind);
        case 3: return tvs_tile_lookup_m3(NW, D, incr, kind);
        case 4: return tvs_tile_lookup_m4(NW, D, incr, kind);
        case 5: return tvs_tile_lookup_m5(NW, D, incr, kind);
        case 6: return tvs_tile_lookup_m6(NW, D, incr, kind);
        default: return nullptr;
    }
}

// Level sets (the option is still called tvs_tile_nw: rounds 2-4 gave the sets to the waves of a workgroup): the fewest whose largest set compiles
// without spills inside the sweep at two wavefronts per SIMD -- every set sweeps the sequences again, and more wavefronts per SIMD buy a
// float64-dense kernel nothing (tvs_tile_kernel.hpp, tvs_waves_per_simd).  A lane's state in doubles, per component: E points of D features + a
// squared norm each, plus the chain value and two previous kernel values for the families that difference kappa along time.  Limits read off the
// compiler's register reports for every built variant: 100 doubles for the linear and the RBF kernel (RBF, M = 4, D = 6: one set of 10 components
// = 100 doubles = 227 registers, no scratch; with increments two sets of 5 = 85 doubles); the families evaluated through base_eval_n at run time
// spill some tens of registers at 90 and are still faster there than the older kernels (Matern-3/2 with increments: 14.7 against 18.8 ms).
int tvs_tile_waves(int M, int D, int E, int kind) {
    const int per_comp = E * (D + 1) + (kind == BASE_LINEAR ? 0 : 3);
    const int limit = (kind == BASE_LINEAR || kind == BASE_RBF) ? 100 : 90;
    for (int NW = 1; NW <= 3; ++NW) {
        if (NW > 1 && M < 3) break;
        if (tvs_max_comps(M, NW) * per_comp <= limit && tvs_tile_lookup(M, NW, D, E == 2, kind)) return NW;
    }
    return 0;
}
}  // namespace gpsig
