// higher-order reverse sweeps, scratch-free, 64 lanes per pair
#define GPSIG_HO_UNDO_ONLY
#define GPSIG_HO_UNDO_G 64
#include "grad_wave_ho_inst.hpp"
