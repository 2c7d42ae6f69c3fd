// tensor-vs-sequence kernels, one lane per sequence: float, num_levels 6
#define TENS_T float
#define TENS_NAME tvs_lookup_f32_m6
#define TENS_MS(X) X(6)
#include "tens_inst_seq.hpp"
