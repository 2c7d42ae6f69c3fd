// higher-order reverse sweeps, scratch-free, 32 lanes per pair
#define GPSIG_HO_UNDO_ONLY
#define GPSIG_HO_UNDO_G 32
#include "grad_wave_ho_inst.hpp"
