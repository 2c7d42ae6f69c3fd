// reverse pass of the tensor-vs-sequence chains, tile kernel, num_levels = 3
#define TVSG_M 3
#include "tvs_grad_tile_inst.hpp"
