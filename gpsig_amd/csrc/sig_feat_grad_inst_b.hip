// Reverse pass of the feature kernels for 5 .. 8 columns (see sig_feat_grad_pick.hpp).
#include "sig_feat_grad_pick.hpp"

namespace gpsig {
SigFeatGradLaunchFn sig_feat_grad_pick_b(int d, int M) {
    switch (d) {
        case 5: return sig_feat_grad_pick<5>(M);
        case 6: return sig_feat_grad_pick<6>(M);
        case 7: return sig_feat_grad_pick<7>(M);
        case 8: return sig_feat_grad_pick<8>(M);
        default: return nullptr;
    }
}
}  // namespace gpsig
