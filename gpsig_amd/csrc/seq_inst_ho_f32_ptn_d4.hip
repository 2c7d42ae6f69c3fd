// float32 higher-order seq-gram kernel instantiations: MODE_PT_NODIFF, D = 4
#define GPSIG_INST_T float
#define GPSIG_INST_NAME seq_lookup_ho_f32_ptn_d4
#define GPSIG_INST_MODE MODE_PT_NODIFF
#define GPSIG_INST_LIST GPSIG_SEQ_HO_D4
#include "seq_inst_ho.hpp"
