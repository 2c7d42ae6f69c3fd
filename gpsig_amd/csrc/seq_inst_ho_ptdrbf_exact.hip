// Exact higher-order seq-gram instances (round 6): the RBF kernel on points with differences, num_levels AND order at compile time, prescaled
// records + table exp -- the shapes of BASELINE configs[1] (16 lanes per pair, 4 columns per lane, 8 / 4 feature columns), order 2, num_levels 4 / 5.
// signature_algs.py:37-74; seq_core.hpp: seq_step_rbf_prescaled_ho.
#include "seq_configs.hpp"
#include "seq_gram_kernel.hpp"

namespace gpsig {
typedef hipError_t (*SeqLaunchFn)(const SeqGramArgs&, int, size_t, hipStream_t);

SeqLaunchFn seq_lookup_ho_ptdrbf_exact(int G, int C, int D, int M, int order) {
#define GPSIG_HO_EXACT(D_, M_, O_) \
    if (G == 16 && C == 4 && D == D_ && M == M_ && order == O_) return &seq_gram_launch<double, 16, 4, D_, M_, MODE_PT_DIFF, true, O_, BASE_RBF>;
    GPSIG_HO_EXACT(8, 4, 2) GPSIG_HO_EXACT(8, 5, 2) GPSIG_HO_EXACT(4, 4, 2) GPSIG_HO_EXACT(4, 5, 2)
    GPSIG_HO_EXACT(8, 3, 2) GPSIG_HO_EXACT(4, 3, 2)
#undef GPSIG_HO_EXACT
    return nullptr;
}
}  // namespace gpsig
