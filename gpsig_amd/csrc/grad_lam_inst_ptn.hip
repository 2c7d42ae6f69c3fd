// seq_lam_undo_kernel instances, MODE_PT_NODIFF
#define GPSIG_INST_LAM
#include "grad_wave_inst.hpp"
namespace gpsig {
Wave2LaunchFn lam_undo_lookup_ptn(int G, int C, int DP, int LQ) { return lam_undo_lookup_mode<MODE_PT_NODIFF>(G, C, DP, LQ); }
}
