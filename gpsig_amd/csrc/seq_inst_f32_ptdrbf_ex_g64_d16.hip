// float32 seq-gram kernel instantiations: MODE_PT_DIFF with the RBF base kernel at compile time, list GPSIG_SEQ_CONFIGS_EX_G64_D16
#define GPSIG_INST_T float
#define GPSIG_INST_NAME seq_lookup_f32_ptdrbf_ex_g64_d16
#define GPSIG_INST_MODE MODE_PT_DIFF
#define GPSIG_INST_KIND BASE_RBF
#define GPSIG_INST_LIST GPSIG_SEQ_CONFIGS_EX_G64_D16
#include "seq_inst.hpp"
