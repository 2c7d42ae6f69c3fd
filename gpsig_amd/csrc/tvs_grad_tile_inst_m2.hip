// reverse pass of the tensor-vs-sequence chains, tile kernel, num_levels = 2
#define TVSG_M 2
#include "tvs_grad_tile_inst.hpp"
