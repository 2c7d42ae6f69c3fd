// seq_grad_fused_kernel instances: 64 lanes per pair, the kernel matrix of the points (difference=False)
#include "grad_fused_inst.hpp"

namespace gpsig {
FusedGradLaunchFn fused_grad_lookup_nodiff_g64(int kind, int DP, int LQ) { return fused_grad_lookup_g<64, false>(kind, DP, LQ); }
}  // namespace gpsig
