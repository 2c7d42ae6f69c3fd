// seq-gram kernel instantiations: MODE_PT_DIFF, SignatureSpectral's kernel at compile time, list GPSIG_SEQ_CONFIGS_SPECTRAL_G64
#define GPSIG_INST_NAME seq_lookup_ptd_spectral_g64
#define GPSIG_INST_MODE MODE_PT_DIFF
#define GPSIG_INST_KIND BASE_SPECTRAL
#define GPSIG_INST_LIST GPSIG_SEQ_CONFIGS_SPECTRAL_G64
#include "seq_inst.hpp"
