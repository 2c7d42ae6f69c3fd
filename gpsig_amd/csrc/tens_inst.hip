// Dispatch over the tens_vs_seq_kernel translation units (tens_inst_{f64,f32}_{lo,m6,m7,m8}.hip; the tensor-lane variant: tens_inst_lanet.hip).
#include "aux_kernels.hpp"

namespace gpsig {
typedef hipError_t (*TvsLaunchFn)(const TvsArgs&, hipStream_t);
#define TENS_DECL(tag) TvsLaunchFn tvs_lookup_##tag##_lo(int, int, bool); TvsLaunchFn tvs_lookup_##tag##_m6(int, int, bool); \
                       TvsLaunchFn tvs_lookup_##tag##_m7(int, int, bool); TvsLaunchFn tvs_lookup_##tag##_m8(int, int, bool);
TENS_DECL(f64) TENS_DECL(f32)
#undef TENS_DECL

TvsLaunchFn tvs_lookup(int M, int TT, bool incr, bool f32) {
    if (M <= 5) return f32 ? tvs_lookup_f32_lo(M, TT, incr) : tvs_lookup_f64_lo(M, TT, incr);
    if (M == 6) return f32 ? tvs_lookup_f32_m6(M, TT, incr) : tvs_lookup_f64_m6(M, TT, incr);
    if (M == 7) return f32 ? tvs_lookup_f32_m7(M, TT, incr) : tvs_lookup_f64_m7(M, TT, incr);
    return f32 ? tvs_lookup_f32_m8(M, TT, incr) : tvs_lookup_f64_m8(M, TT, incr);
}
}  // namespace gpsig
