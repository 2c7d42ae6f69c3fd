// float32 seq-gram kernel instantiations: MODE_PT_DIFF, list GPSIG_SEQ_CONFIGS_F32_G64
#define GPSIG_INST_T float
#define GPSIG_INST_NAME seq_lookup_f32_ptd_g64
#define GPSIG_INST_MODE MODE_PT_DIFF
#define GPSIG_INST_LIST GPSIG_SEQ_CONFIGS_F32_G64
#include "seq_inst.hpp"
