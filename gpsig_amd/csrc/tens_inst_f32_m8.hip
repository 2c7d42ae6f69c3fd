// tensor-vs-sequence kernels, one lane per sequence: float, num_levels 8
#define TENS_T float
#define TENS_NAME tvs_lookup_f32_m8
#define TENS_MS(X) X(8)
#include "tens_inst_seq.hpp"
