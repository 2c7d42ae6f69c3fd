// Reverse pass of the feature kernels for 9 .. 12 columns (see sig_feat_grad_pick.hpp).
#include "sig_feat_grad_pick.hpp"

namespace gpsig {
SigFeatGradLaunchFn sig_feat_grad_pick_c(int d, int M) {
    switch (d) {
        case 9: return sig_feat_grad_pick<9>(M);
        case 10: return sig_feat_grad_pick<10>(M);
        case 11: return sig_feat_grad_pick<11>(M);
        case 12: return sig_feat_grad_pick<12>(M);
        default: return nullptr;
    }
}
}  // namespace gpsig
