// higher-order reverse sweeps, scratch-free, 16 lanes per pair
#define GPSIG_HO_UNDO_ONLY
#define GPSIG_HO_UNDO_G 16
#include "grad_wave_ho_inst.hpp"
