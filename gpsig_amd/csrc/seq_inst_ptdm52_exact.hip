// seq-gram kernel instantiations: MODE_PT_DIFF with a Matern base kernel at compile time (BASE_MATERN52), list GPSIG_SEQ_CONFIGS_EXACT
#define GPSIG_INST_NAME seq_lookup_ptdm52_exact
#define GPSIG_INST_MODE MODE_PT_DIFF
#define GPSIG_INST_KIND BASE_MATERN52
#define GPSIG_INST_LIST GPSIG_SEQ_CONFIGS_EXACT
#include "seq_inst.hpp"
