// seq_grad_wave_kernel instances, MODE_INC
#include "grad_wave_inst.hpp"
namespace gpsig {
WaveLaunchFn wave_lookup_inc(int G, int C, int DP, int LQ) { return wave_lookup_mode<MODE_INC>(G, C, DP, LQ); }
}
