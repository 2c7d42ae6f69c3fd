// seq_inst_ho.hpp -- explicit instantiation helper for the higher-order kernels (see seq_inst.hpp).
#include "seq_configs.hpp"
#include "seq_gram_kernel.hpp"

#ifndef GPSIG_INST_T
#define GPSIG_INST_T double
#endif

namespace gpsig {
typedef hipError_t (*SeqLaunchFn)(const SeqGramArgs&, int, size_t, hipStream_t);

#define GPSIG_INST_HO_CASE(G_, C_, D_, MM_, OM_) \
    if (G == G_ && C == C_ && D == D_ && MMAX == MM_ && OMAX == OM_) \
        return &seq_gram_launch<GPSIG_INST_T, G_, C_, D_, MM_, GPSIG_INST_MODE, false, OM_>;

SeqLaunchFn GPSIG_INST_NAME(int G, int C, int D, int MMAX, int OMAX) {
    GPSIG_INST_LIST(GPSIG_INST_HO_CASE)
    return nullptr;
}
}  // namespace gpsig
