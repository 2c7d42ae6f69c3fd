// tensor-vs-sequence tile kernel, num_levels = 5
#define TVS_TILE_M 5
#define TVS_TILE_NWS(X) X(2) X(3)
#include "tvs_tile_inst.hpp"
