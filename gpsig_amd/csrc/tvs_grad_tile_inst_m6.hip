// reverse pass of the tensor-vs-sequence chains, tile kernel, num_levels = 6
#define TVSG_M 6
#include "tvs_grad_tile_inst.hpp"
