// reverse pass of the tensor-vs-sequence chains, tile kernel, num_levels = 1
#define TVSG_M 1
#include "tvs_grad_tile_inst.hpp"
