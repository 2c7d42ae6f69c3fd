// Feature kernels for 1 .. 4 columns (see sig_feat_pick.hpp).
#include "sig_feat_pick.hpp"

namespace gpsig {
SigFeatLaunchFn sig_feat_pick_a(int d, int M) {
    switch (d) {
        case 1: return sig_feat_pick<1>(M);
        case 2: return sig_feat_pick<2>(M);
        case 3: return sig_feat_pick<3>(M);
        case 4: return sig_feat_pick<4>(M);
        default: return nullptr;
    }
}
}  // namespace gpsig
