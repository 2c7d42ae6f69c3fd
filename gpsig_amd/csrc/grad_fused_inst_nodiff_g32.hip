// seq_grad_fused_kernel instances: 32 lanes per pair (two pairs per wavefront), the kernel matrix of the points (difference=False)
#include "grad_fused_inst.hpp"

namespace gpsig {
FusedGradLaunchFn fused_grad_lookup_nodiff_g32(int kind, int DP, int LQ) { return fused_grad_lookup_g<32, false>(kind, DP, LQ); }
}  // namespace gpsig
