// seq-gram kernel instantiations: the float64 RBF instances of the fused reverse kernel's headline shapes (16 lanes per pair, 4 columns per lane,
// 4 / 8 padded features, num_levels 4 / 5 at compile time) that also write the reverse pass's stash (seq_gram_kernel.hpp: STASH)
#include "seq_configs.hpp"
#include "seq_gram_kernel.hpp"

namespace gpsig {
typedef hipError_t (*SeqLaunchFn)(const SeqGramArgs&, int, size_t, hipStream_t);

SeqLaunchFn seq_lookup_ptdrbf_stash(int G, int C, int D, int MMAX) {
    if (G != 16 || C != 4) return nullptr;
#define ST(D_, M_) if (D == D_ && MMAX == M_) return &seq_gram_launch<double, 16, 4, D_, M_, MODE_PT_DIFF, true, 0, BASE_RBF, true>;
    ST(4, 4) ST(4, 5) ST(8, 4) ST(8, 5)
#undef ST
    return nullptr;
}

// the same shapes with a Matern family at compile time
SeqLaunchFn seq_lookup_ptdmatern_stash(int kind, int G, int C, int D, int MMAX) {
    if (G != 16 || C != 4) return nullptr;
#define ST(K_, D_, M_) if (kind == K_ && D == D_ && MMAX == M_) return &seq_gram_launch<double, 16, 4, D_, M_, MODE_PT_DIFF, true, 0, K_, true>;
#define STK(K_) ST(K_, 4, 4) ST(K_, 4, 5) ST(K_, 8, 4) ST(K_, 8, 5)
    STK(BASE_MATERN12) STK(BASE_MATERN32) STK(BASE_MATERN52)
#undef STK
#undef ST
    return nullptr;
}
}  // namespace gpsig
