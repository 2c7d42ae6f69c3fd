// seq_inst.hpp -- explicit instantiation helper: each seq_inst_*.hip defines GPSIG_INST_NAME, GPSIG_INST_MODE and
// GPSIG_INST_LIST and includes this file, which emits the kernels of that list and a lookup function.
#include "seq_configs.hpp"
#include "seq_gram_kernel.hpp"

#ifndef GPSIG_INST_T
#define GPSIG_INST_T double
#endif
#ifndef GPSIG_INST_KIND
#define GPSIG_INST_KIND -1
#endif

namespace gpsig {
typedef hipError_t (*SeqLaunchFn)(const SeqGramArgs&, int, size_t, hipStream_t);

#define GPSIG_INST_CASE(G_, C_, D_, MM_, EX_) \
    if (G == G_ && C == C_ && D == D_ && MMAX == MM_ && exact == EX_) \
        return &seq_gram_launch<GPSIG_INST_T, G_, C_, D_, MM_, GPSIG_INST_MODE, EX_, 0, GPSIG_INST_KIND>;

SeqLaunchFn GPSIG_INST_NAME(int G, int C, int D, int MMAX, bool exact) {
    GPSIG_INST_LIST(GPSIG_INST_CASE)
    return nullptr;
}
}  // namespace gpsig
