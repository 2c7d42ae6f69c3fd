// Feature kernels for 13 .. 16 columns (see sig_feat_pick.hpp).
#include "sig_feat_pick.hpp"

namespace gpsig {
SigFeatLaunchFn sig_feat_pick_d(int d, int M) {
    switch (d) {
        case 13: return sig_feat_pick<13>(M);
        case 14: return sig_feat_pick<14>(M);
        case 15: return sig_feat_pick<15>(M);
        case 16: return sig_feat_pick<16>(M);
        default: return nullptr;
    }
}
}  // namespace gpsig
