// seq_lam_undo_kernel instances, MODE_PT_NODIFF, the RBF base kernel at compile time
#define GPSIG_INST_LAM
#include "grad_wave_inst.hpp"
namespace gpsig {
Wave2LaunchFn lam_undo_lookup_ptn_rbf(int G, int C, int DP, int LQ) { return LamUndoInst<BASE_RBF>::lookup<MODE_PT_NODIFF>(G, C, DP, LQ); }
}
