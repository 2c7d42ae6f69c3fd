// Reverse pass of the feature kernels for 13 .. 16 columns (see sig_feat_grad_pick.hpp).
#include "sig_feat_grad_pick.hpp"

namespace gpsig {
SigFeatGradLaunchFn sig_feat_grad_pick_d(int d, int M) {
    switch (d) {
        case 13: return sig_feat_grad_pick<13>(M);
        case 14: return sig_feat_grad_pick<14>(M);
        case 15: return sig_feat_grad_pick<15>(M);
        case 16: return sig_feat_grad_pick<16>(M);
        default: return nullptr;
    }
}
}  // namespace gpsig
