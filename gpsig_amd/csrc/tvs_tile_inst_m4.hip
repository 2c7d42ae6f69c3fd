// tensor-vs-sequence tile kernel, num_levels = 4
#define TVS_TILE_M 4
#define TVS_TILE_NWS(X) X(1) X(2)
#include "tvs_tile_inst.hpp"
