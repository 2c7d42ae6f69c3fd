// Host check of the level partition of the Kzx tile kernel (gpsig_amd/csrc/tvs_plan.hpp: the constexpr helpers the device code
// uses as template arguments): every level on exactly one wave, local offsets consistent, the balance the planner relies on.
#include <cstdio>
#include "tvs_plan.hpp"

int main() {
    int bad = 0;
    for (int M = 1; M <= 8; ++M)
        for (int NW = 1; NW <= 3; ++NW) {
            int seen = 0, total = 0, worst = 0;
            for (int w = 0; w < NW; ++w) {
                const int mask = gpsig::tvs_level_mask(M, NW, w);
                if (mask & seen) ++bad;                              // a level on two waves
                if (mask & 1) ++bad;                                 // level 0 is nobody's chain
                seen |= mask;
                const int c = gpsig::tvs_mask_comps(mask);
                total += c;
                if (c > worst) worst = c;
                int off = 0;                                         // local offsets are the running sum of the mask's levels
                for (int i = 1; i <= M; ++i)
                    if ((mask >> i) & 1) { if (gpsig::tvs_local_off(mask, i) != off) ++bad; off += i; }
            }
            if (seen != ((1 << (M + 1)) - 2)) ++bad;                 // every level 1..M exactly once
            if (total != M * (M + 1) / 2) ++bad;
            if (worst != gpsig::tvs_max_comps(M, NW)) ++bad;
            // longest-processing-time assignment: no wave carries more than the even share plus the largest level
            if (worst > (M * (M + 1) / 2 + NW - 1) / NW + M) ++bad;
        }
    // the shapes the design document quotes
    if (gpsig::tvs_max_comps(4, 2) != 5 || gpsig::tvs_max_comps(5, 3) != 5 || gpsig::tvs_max_comps(6, 3) != 7) ++bad;
    printf("%d\n", bad);
    return bad != 0;
}
