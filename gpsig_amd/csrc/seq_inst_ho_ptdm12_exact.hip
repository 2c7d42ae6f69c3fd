// Exact higher-order seq-gram instances for the Matern families (round 6, as seq_inst_ho_ptdrbf_exact.hip for the RBF kernel): points with differences,
// num_levels AND order at compile time, prescaled records + table exp + v_rsq_f64 (seq_core.hpp: seq_step_matern_prescaled_ho); 16 lanes per pair, 4 columns per
// lane, 8 / 4 feature columns, order 2, num_levels 3 / 4 / 5.  signature_algs.py:37-74.
#include "seq_configs.hpp"
#include "seq_gram_kernel.hpp"

namespace gpsig {
typedef hipError_t (*SeqLaunchFn)(const SeqGramArgs&, int, size_t, hipStream_t);

SeqLaunchFn seq_lookup_ho_ptdm12_exact(int kind, int G, int C, int D, int M, int order) {
#define GPSIG_HO_EXACT_M(K_, D_, M_, O_) \
    if (kind == K_ && G == 16 && C == 4 && D == D_ && M == M_ && order == O_) return &seq_gram_launch<double, 16, 4, D_, M_, MODE_PT_DIFF, true, O_, K_>;
#define GPSIG_HO_EXACT_K(K_) \
    GPSIG_HO_EXACT_M(K_, 8, 4, 2) GPSIG_HO_EXACT_M(K_, 8, 5, 2) GPSIG_HO_EXACT_M(K_, 4, 4, 2) GPSIG_HO_EXACT_M(K_, 4, 5, 2) GPSIG_HO_EXACT_M(K_, 8, 3, 2) GPSIG_HO_EXACT_M(K_, 4, 3, 2)
    GPSIG_HO_EXACT_K(BASE_MATERN12)
#undef GPSIG_HO_EXACT_K
#undef GPSIG_HO_EXACT_M
    return nullptr;
}
}  // namespace gpsig
