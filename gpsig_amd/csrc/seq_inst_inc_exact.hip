// seq-gram kernel instantiations: MODE_INC, list GPSIG_SEQ_CONFIGS_EXACT
#define GPSIG_INST_NAME seq_lookup_inc_exact
#define GPSIG_INST_MODE MODE_INC
#define GPSIG_INST_LIST GPSIG_SEQ_CONFIGS_EXACT
#include "seq_inst.hpp"
