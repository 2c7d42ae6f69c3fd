// seq-gram kernel instantiations: MODE_PT_DIFF with the RBF base kernel at compile time, list GPSIG_SEQ_CONFIGS_EXACT
#define GPSIG_INST_NAME seq_lookup_ptdrbf_exact
#define GPSIG_INST_MODE MODE_PT_DIFF
#define GPSIG_INST_KIND BASE_RBF
#define GPSIG_INST_LIST GPSIG_SEQ_CONFIGS_EXACT
#include "seq_inst.hpp"
