// seq_grad_fused_kernel instances: 16 lanes per pair, double increments (difference=True)
#include "grad_fused_inst.hpp"

namespace gpsig {
FusedGradLaunchFn fused_grad_lookup_diff_g16(int kind, int DP, int LQ) { return fused_grad_lookup_g<16, true>(kind, DP, LQ); }
}  // namespace gpsig
