// instances of seq_grad_wave_ho_kernel (grad_wave_ho_kernel.hpp): lanes per pair x columns per lane x order (2, 3, 4), num_levels <= 5
#include "grad_wave_ho_kernel.hpp"

namespace gpsig {
typedef hipError_t (*WaveHoLaunchFn)(const WaveHoArgs&, int, hipStream_t);

template <int G, int C, int O>
static hipError_t wave_ho_launch(const WaveHoArgs& a, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL((seq_grad_wave_ho_kernel<G, C, 4, O>), dim3(nblocks), dim3(64), 0, s, a);
    return hipGetLastError();
}

template <int G, int C>
static WaveHoLaunchFn pick(int order) {
    if (order == 2) return &wave_ho_launch<G, C, 2>;
    if (order == 3) return &wave_ho_launch<G, C, 3>;
    if (order == 4) return &wave_ho_launch<G, C, 4>;
    return nullptr;
}

// order: min(order, num_levels) >= 2
WaveHoLaunchFn wave_ho_lookup(int G, int C, int order, int M) {
    if (M < 2 || M > 5) return nullptr;
    if (G == 16 && C == 2) return pick<16, 2>(order);
    if (G == 16 && C == 4) return pick<16, 4>(order);
    if (G == 64 && C == 2) return pick<64, 2>(order);
    if (G == 64 && C == 4) return pick<64, 4>(order);
    if (G == 64 && C == 8) return pick<64, 8>(order);
    return nullptr;
}
}  // namespace gpsig
