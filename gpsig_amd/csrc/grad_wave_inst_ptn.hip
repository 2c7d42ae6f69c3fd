// seq_grad_wave_kernel instances, MODE_PT_NODIFF
#include "grad_wave_inst.hpp"
namespace gpsig {
WaveLaunchFn wave_lookup_ptn(int G, int C, int DP, int LQ) { return wave_lookup_mode<MODE_PT_NODIFF>(G, C, DP, LQ); }
}
