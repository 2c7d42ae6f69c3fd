// seq_grad_fused_kernel instances: 16 lanes per pair, the kernel matrix of the points (difference=False)
#include "grad_fused_inst.hpp"

namespace gpsig {
FusedGradLaunchFn fused_grad_lookup_nodiff_g16(int kind, int DP, int LQ) { return fused_grad_lookup_g<16, false>(kind, DP, LQ); }
}  // namespace gpsig
