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

// Waves per workgroup.  A lane's state in doubles, per component: E points of D features + a squared norm each, plus the chain value and two
// previous kernel values for the families that difference kappa along time (tvs_state_doubles).  First choice (round 5): the fewest waves whose
// largest level set keeps the instance at 168 registers = THREE wavefronts per SIMD (state <= 70 doubles; those instances are compiled with that
// launch bound) -- a gfx950 SIMD needs three wavefronts to issue a float64 instruction every 4 cycles, two get one every 5.3.  BASELINE configs[2]
// with incremental tensors (M = 4, D = 6, E = 2): two waves of 5 components = 85 doubles = 236 registers, three waves of 4 / 3 / 3 = 68 doubles.
// Otherwise the fewest waves that compile without spills at two wavefronts per SIMD, limits read off the compiler's register reports for every
// built variant: 100 doubles for the linear kernel, 90 for RBF; the families evaluated through base_eval_n at run time spill some tens of registers
// at 90 and are still faster there than the older kernels (Matern-3/2 with increments: 14.7 against 18.8 ms).
int tvs_tile_waves(int M, int D, int E, int kind) {
    const bool incr = E == 2;
    for (int NW = 1; NW <= 3; ++NW) {
        if (NW > 1 && M < 3) break;
        if (kind != BASE_LINEAR && tvs_waves_per_simd(M, NW, D, incr, kind) == 3 && tvs_tile_lookup(M, NW, D, incr, kind)) return NW;
    }
    const int per_comp = E * (D + 1) + (kind == BASE_LINEAR ? 0 : 3);
    const int limit = kind == BASE_LINEAR ? 100 : 90;
    for (int NW = 1; NW <= 3; ++NW) {
        if (NW > 1 && M < 3) break;
        if (tvs_max_comps(M, NW) * per_comp <= limit && tvs_tile_lookup(M, NW, D, E == 2, kind)) return NW;
    }
    return 0;
}
}  // namespace gpsig
