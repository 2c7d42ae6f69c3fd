// tensor-vs-sequence kernels, one lane per sequence: double, num_levels 7
#define TENS_T double
#define TENS_NAME tvs_lookup_f64_m7
#define TENS_MS(X) X(7)
#include "tens_inst_seq.hpp"
