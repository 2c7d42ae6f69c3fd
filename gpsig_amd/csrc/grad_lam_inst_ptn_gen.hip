// seq_lam_undo_kernel instances, MODE_PT_NODIFF, base kernel at run time
#define GPSIG_INST_LAM
#include "grad_wave_inst.hpp"
namespace gpsig {
Wave2LaunchFn lam_undo_lookup_ptn_gen(int G, int C, int DP, int LQ) { return LamUndoInst<-1>::lookup<MODE_PT_NODIFF>(G, C, DP, LQ); }
}
