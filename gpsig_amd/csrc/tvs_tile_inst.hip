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

// Level sets (the option is still called tvs_tile_nw: rounds 2-4 gave the sets to the waves of a workgroup): tvs_planned_sets of
// tvs_tile_kernel.hpp -- the fewest whose largest set compiles without spills inside the sweep at two wavefronts per SIMD: every set sweeps the
// sequences again, and more wavefronts per SIMD buy a float64-dense kernel nothing.  kind: BASE_LINEAR, BASE_RBF, a Matern family, or -1.
int tvs_tile_waves(int M, int D, int E, int kind) {
    if (M < 2 || M > 6) return 0;
    int P = 0;
    switch (M * 16 + D) {
#define TVS_PLAN_CASE(M_, D_) case M_ * 16 + D_: \
        P = kind == BASE_LINEAR ? tvs_planned_sets(M_, D_, false, BASE_LINEAR) : \
            kind == BASE_RBF ? (E == 2 ? tvs_planned_sets(M_, D_, true, BASE_RBF) : tvs_planned_sets(M_, D_, false, BASE_RBF)) : \
            tvs_is_matern(kind) ? (E == 2 ? tvs_planned_sets(M_, D_, true, BASE_MATERN32) : tvs_planned_sets(M_, D_, false, BASE_MATERN32)) : \
            (E == 2 ? tvs_planned_sets(M_, D_, true, -1) : tvs_planned_sets(M_, D_, false, -1)); break;
        TVS_PLAN_CASE(2, 4) TVS_PLAN_CASE(2, 6) TVS_PLAN_CASE(2, 8) TVS_PLAN_CASE(3, 4) TVS_PLAN_CASE(3, 6) TVS_PLAN_CASE(3, 8)
        TVS_PLAN_CASE(4, 4) TVS_PLAN_CASE(4, 6) TVS_PLAN_CASE(4, 8) TVS_PLAN_CASE(5, 4) TVS_PLAN_CASE(5, 6) TVS_PLAN_CASE(5, 8)
        TVS_PLAN_CASE(6, 4) TVS_PLAN_CASE(6, 6) TVS_PLAN_CASE(6, 8)
#undef TVS_PLAN_CASE
        default: return 0;
    }
    return (P > 0 && tvs_tile_lookup(M, P, D, E == 2, kind)) ? P : 0;
}
}  // namespace gpsig
