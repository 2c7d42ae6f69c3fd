// tensor-vs-sequence tile kernel, num_levels = 2
#define TVS_TILE_M 2
#define TVS_TILE_NWS(X) X(1)
#include "tvs_tile_inst.hpp"
