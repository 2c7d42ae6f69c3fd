// higher-order reverse sweeps: the slot kernel (prefixes through HBM)
#include "grad_wave_ho_inst.hpp"
