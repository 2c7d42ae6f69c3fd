// sig_pieces.hpp -- how the feature contraction (sig_feat_kernel.hpp) cuts its depth into pieces.  Plain C++ (no HIP): compiled on the
// host by tests/test_device_headers.py as well.
#pragma once

#include <stdint.h>

namespace gpsig {

// Slab ranges of the depth pieces.  `equal` pieces of the same size, the last of which is cut into `graded` finer ones of halving size
// (1/2, 1/4, .., the last two alike) when graded > 1: workgroups are handed out piece after piece, so equal pieces end in a last round
// that is as long as the others but only partly full (configs[1]: 528 tiles x 11 pieces on 512 workgroup slots = 11.34 rounds, paid as
// 12), while finer pieces at the end fill the slots that fall free there, and the launch ends within one SMALL piece of
// (total work / slots).  Returns the number of pieces.  Depends on the depth and the two counts alone (api.hip: chosen from the full
// problem's size), so an entry's summation order is the same in every tile, row block and rank.
inline int sig_piece_bounds(int nslab, int equal, int graded, int* bound) {
    if (equal < 1) equal = 1;
    if (equal > nslab) equal = nslab < 1 ? 1 : nslab;
    int n = 0;
    for (int s = 0; s < equal - 1; ++s) bound[n++] = int(int64_t(nslab) * s / equal);
    const int last0 = int(int64_t(nslab) * (equal - 1) / equal), len = nslab - last0;
    bound[n++] = last0;
    if (graded > 1 && len >= 2 * graded) {
        int at = last0, left = len;
        for (int g = 1; g < graded; ++g) {         // 1/2, 1/4, ... of the last piece; the final one takes what is left
            const int take = left / 2 > 0 ? left / 2 : 1;
            at += take; left -= take;
            bound[n++] = at;
        }
    }
    bound[n] = nslab;
    return n;
}

}  // namespace gpsig
