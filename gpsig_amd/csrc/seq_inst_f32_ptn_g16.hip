// float32 seq-gram kernel instantiations: MODE_PT_NODIFF, list GPSIG_SEQ_CONFIGS_F32_G16
#define GPSIG_INST_T float
#define GPSIG_INST_NAME seq_lookup_f32_ptn_g16
#define GPSIG_INST_MODE MODE_PT_NODIFF
#define GPSIG_INST_LIST GPSIG_SEQ_CONFIGS_F32_G16
#include "seq_inst.hpp"
