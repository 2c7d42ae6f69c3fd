// tvs_plan.hpp -- how the Kzx tile kernel (tvs_tile_kernel.hpp) splits the signature levels over the wavefronts of a workgroup.
// HIP-free constexpr helpers: used as template arguments on the device, by the host planner, and by tests/emu/test_tvs_plan.cpp.
#pragma once

namespace gpsig {

// levels of wave w out of NW: longest-processing-time assignment of levels M, M-1, .. 1 (level i costs i components)
constexpr int tvs_level_mask(int M, int NW, int w) {
    int load[4] = {0, 0, 0, 0}, mask[4] = {0, 0, 0, 0};
    for (int i = M; i >= 1; --i) {
        int best = 0;
        for (int k = 1; k < NW; ++k)
            if (load[k] < load[best]) best = k;
        load[best] += i;
        mask[best] |= 1 << i;
    }
    return mask[w];
}
constexpr int tvs_mask_comps(int mask) {
    int n = 0;
    for (int i = 1; i < 16; ++i)
        if ((mask >> i) & 1) n += i;
    return n;
}
constexpr int tvs_max_comps(int M, int NW) {
    int b = 0;
    for (int w = 0; w < NW; ++w) {
        const int c = tvs_mask_comps(tvs_level_mask(M, NW, w));
        if (c > b) b = c;
    }
    return b;
}
// first local component of level i within the mask
constexpr int tvs_local_off(int mask, int i) {
    int n = 0;
    for (int l = 1; l < i; ++l)
        if ((mask >> l) & 1) n += l;
    return n;
}

}  // namespace gpsig
