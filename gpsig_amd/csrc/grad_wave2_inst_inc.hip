// seq_grad_wave2_kernel instances, MODE_INC (the linear kernel on increments)
#define GPSIG_INST_WAVE2
#include "grad_wave_inst.hpp"
namespace gpsig {
Wave2LaunchFn wave2_lookup_inc(int G, int C, int DP, int LQ) { return wave2_lookup_mode<MODE_INC>(G, C, DP, LQ); }
}
