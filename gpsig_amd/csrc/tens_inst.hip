// tensor-vs-sequence kernel instantiations: num_levels 1..8, 1 or 2 tensors per wave, with / without increments
#include "aux_kernels.hpp"

namespace gpsig {
typedef hipError_t (*TvsLaunchFn)(const TvsArgs&, hipStream_t);

template <int M, int TT, bool INCR>
static hipError_t tvs_launch(const TvsArgs& A, hipStream_t stream) {
    dim3 grid((unsigned)((A.N + 63) / 64), (unsigned)((A.Tn + TT - 1) / TT));
    hipLaunchKernelGGL((tens_vs_seq_kernel<double, M, TT, INCR>), grid, dim3(64), 0, stream, A);
    return hipGetLastError();
}

#define TVS_CASE(M_)                                                              \
    if (M == M_) {                                                                \
        if (TT == 1) return incr ? &tvs_launch<M_, 1, true> : &tvs_launch<M_, 1, false>; \
        return incr ? &tvs_launch<M_, 2, true> : &tvs_launch<M_, 2, false>;       \
    }

TvsLaunchFn tvs_lookup(int M, int TT, bool incr) {
    TVS_CASE(1) TVS_CASE(2) TVS_CASE(3) TVS_CASE(4) TVS_CASE(5) TVS_CASE(6) TVS_CASE(7) TVS_CASE(8)
    return nullptr;
}
}  // namespace gpsig
