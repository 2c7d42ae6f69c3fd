// seq_grad_wave_kernel instances, MODE_PT_DIFF
#include "grad_wave_inst.hpp"
namespace gpsig {
WaveLaunchFn wave_lookup_ptd(int G, int C, int DP, int LQ) { return wave_lookup_mode<MODE_PT_DIFF>(G, C, DP, LQ); }
}
