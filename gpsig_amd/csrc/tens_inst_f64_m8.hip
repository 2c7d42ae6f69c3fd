// tensor-vs-sequence kernels, one lane per sequence: double, num_levels 8
#define TENS_T double
#define TENS_NAME tvs_lookup_f64_m8
#define TENS_MS(X) X(8)
#include "tens_inst_seq.hpp"
