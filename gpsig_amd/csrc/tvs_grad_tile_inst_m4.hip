// reverse pass of the tensor-vs-sequence chains, tile kernel, num_levels = 4
#define TVSG_M 4
#include "tvs_grad_tile_inst.hpp"
