// tens_inst_seq.hpp -- one translation unit of tens_vs_seq_kernel instantiations (one lane per sequence; 1 or 2 tensors per wave, with / without
// increments): #define TENS_T (element type), TENS_NAME (the unit's lookup function) and TENS_MS(X) (its num_levels values) before including.
// (Round 5: all of num_levels 1..8 in both precisions used to be ONE unit, which took seven minutes by itself -- the longest pole of the build.)
#include "aux_kernels.hpp"

namespace gpsig {
typedef hipError_t (*TvsLaunchFn)(const TvsArgs&, hipStream_t);

template <typename T, int M, int TT, bool INCR>
static hipError_t tvs_launch(const TvsArgs& A, hipStream_t stream) {
    dim3 grid((unsigned)((A.N + 63) / 64), (unsigned)((A.Tn + TT - 1) / TT));
    hipLaunchKernelGGL((tens_vs_seq_kernel<T, M, TT, INCR>), grid, dim3(64), 0, stream, A);
    return hipGetLastError();
}

// Two tensors per wave are built where the host can ask for them (api.hip: lt (2 + E) 2 <= 100 doubles of lane state, i.e. num_levels <= 4 with
// increments, <= 5 without): tens_vs_seq_kernel<float, 8, 2, true> alone took 6.6 minutes to compile and was never launched.
template <typename T, int M, bool INCR>
static TvsLaunchFn tvs_two(void) {
    if constexpr ((M * (M + 1) / 2) * (2 + (INCR ? 2 : 1)) * 2 <= 100) return &tvs_launch<T, M, 2, INCR>;
    else return nullptr;
}

TvsLaunchFn TENS_NAME(int M, int TT, bool incr) {
    typedef TENS_T T;
#define TVS_CASE(M_)                                                              \
    if (M == M_) {                                                                \
        if (TT == 1) return incr ? &tvs_launch<T, M_, 1, true> : &tvs_launch<T, M_, 1, false>; \
        return incr ? tvs_two<T, M_, true>() : tvs_two<T, M_, false>();          \
    }
    TENS_MS(TVS_CASE)
#undef TVS_CASE
    return nullptr;
}
}  // namespace gpsig
