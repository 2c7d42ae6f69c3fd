// seq_lam_undo_kernel instances, MODE_PT_DIFF
#define GPSIG_INST_LAM
#include "grad_wave_inst.hpp"
namespace gpsig {
Wave2LaunchFn lam_undo_lookup_ptd(int G, int C, int DP, int LQ) { return lam_undo_lookup_mode<MODE_PT_DIFF>(G, C, DP, LQ); }
}
