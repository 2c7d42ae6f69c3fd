// tensor-vs-sequence kernels, one lane per sequence: float, num_levels 1 2 3 4 5
#define TENS_T float
#define TENS_NAME tvs_lookup_f32_lo
#define TENS_MS(X) X(1) X(2) X(3) X(4) X(5)
#include "tens_inst_seq.hpp"
