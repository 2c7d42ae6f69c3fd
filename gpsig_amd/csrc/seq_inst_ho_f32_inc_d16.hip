// float32 higher-order seq-gram kernel instantiations: MODE_INC, D = 16
#define GPSIG_INST_T float
#define GPSIG_INST_NAME seq_lookup_ho_f32_inc_d16
#define GPSIG_INST_MODE MODE_INC
#define GPSIG_INST_LIST GPSIG_SEQ_HO_D16
#include "seq_inst_ho.hpp"
