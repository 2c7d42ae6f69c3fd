// reverse pass of the tensor-vs-sequence chains, tile kernel, num_levels = 5
#define TVSG_M 5
#include "tvs_grad_tile_inst.hpp"
