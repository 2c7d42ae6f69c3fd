// tensor-vs-sequence kernels, one lane per sequence: double, num_levels 6
#define TENS_T double
#define TENS_NAME tvs_lookup_f64_m6
#define TENS_MS(X) X(6)
#include "tens_inst_seq.hpp"
