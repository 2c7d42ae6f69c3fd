// higher-order seq-gram kernel instantiations: MODE_PT_DIFF, D = 32, float32
#define GPSIG_INST_T float
#define GPSIG_INST_NAME seq_lookup_ho_f32_ptd_d32
#define GPSIG_INST_MODE MODE_PT_DIFF
#define GPSIG_INST_LIST GPSIG_SEQ_HO_D32
#include "seq_inst_ho.hpp"
