// Exact higher-order seq-gram instances, orders 3 and 4 (round 6): as seq_inst_ho_ptdrbf_exact.hip on the shape the planner takes for up to 128 record
// rows at these orders (64 lanes per pair, 2 columns per lane), 8 feature columns, num_levels 4 / 5.
#include "seq_configs.hpp"
#include "seq_gram_kernel.hpp"

namespace gpsig {
typedef hipError_t (*SeqLaunchFn)(const SeqGramArgs&, int, size_t, hipStream_t);

SeqLaunchFn seq_lookup_ho_ptdrbf_exact_o4(int G, int C, int D, int M, int order) {
#define GPSIG_HO_EXACT(D_, M_, O_) \
    if (G == 64 && C == 2 && D == D_ && M == M_ && order == O_) return &seq_gram_launch<double, 64, 2, D_, M_, MODE_PT_DIFF, true, O_, BASE_RBF>;
    GPSIG_HO_EXACT(8, 4, 3) GPSIG_HO_EXACT(8, 5, 3) GPSIG_HO_EXACT(8, 5, 4) GPSIG_HO_EXACT(8, 4, 4)
#undef GPSIG_HO_EXACT
    return nullptr;
}
}  // namespace gpsig
