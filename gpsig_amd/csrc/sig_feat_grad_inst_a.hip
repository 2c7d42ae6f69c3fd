// Reverse pass of the feature kernels for 1 .. 4 columns (see sig_feat_grad_pick.hpp).
#include "sig_feat_grad_pick.hpp"

namespace gpsig {
SigFeatGradLaunchFn sig_feat_grad_pick_a(int d, int M) {
    switch (d) {
        case 1: return sig_feat_grad_pick<1>(M);
        case 2: return sig_feat_grad_pick<2>(M);
        case 3: return sig_feat_grad_pick<3>(M);
        case 4: return sig_feat_grad_pick<4>(M);
        default: return nullptr;
    }
}
}  // namespace gpsig
