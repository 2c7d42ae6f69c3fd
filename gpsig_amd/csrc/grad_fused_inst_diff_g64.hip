// seq_grad_fused_kernel instances: 64 lanes per pair, double increments (difference=True)
#include "grad_fused_inst.hpp"

namespace gpsig {
FusedGradLaunchFn fused_grad_lookup_diff_g64(int kind, int DP, int LQ) { return fused_grad_lookup_g<64, true>(kind, DP, LQ); }
}  // namespace gpsig
