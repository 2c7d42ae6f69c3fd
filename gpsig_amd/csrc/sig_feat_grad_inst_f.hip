// Reverse pass of the feature kernels for 25 .. 32 columns (see sig_feat_grad_pick.hpp).
#include "sig_feat_grad_pick.hpp"

namespace gpsig {
SigFeatGradLaunchFn sig_feat_grad_pick_f(int d, int M) {
    switch (d) {
        case 25: return sig_feat_grad_pick<25>(M);
        case 26: return sig_feat_grad_pick<26>(M);
        case 27: return sig_feat_grad_pick<27>(M);
        case 28: return sig_feat_grad_pick<28>(M);
        case 29: return sig_feat_grad_pick<29>(M);
        case 30: return sig_feat_grad_pick<30>(M);
        case 31: return sig_feat_grad_pick<31>(M);
        case 32: return sig_feat_grad_pick<32>(M);
        default: return nullptr;
    }
}
}  // namespace gpsig
