// Reverse pass of the feature kernels for 17 .. 24 columns (see sig_feat_grad_pick.hpp).
#include "sig_feat_grad_pick.hpp"

namespace gpsig {
SigFeatGradLaunchFn sig_feat_grad_pick_e(int d, int M) {
    switch (d) {
        case 17: return sig_feat_grad_pick<17>(M);
        case 18: return sig_feat_grad_pick<18>(M);
        case 19: return sig_feat_grad_pick<19>(M);
        case 20: return sig_feat_grad_pick<20>(M);
        case 21: return sig_feat_grad_pick<21>(M);
        case 22: return sig_feat_grad_pick<22>(M);
        case 23: return sig_feat_grad_pick<23>(M);
        case 24: return sig_feat_grad_pick<24>(M);
        default: return nullptr;
    }
}
}  // namespace gpsig
