// instances of the higher-order reverse sweeps (grad_wave_ho_kernel.hpp): lanes per pair x columns per lane x order (2, 3, 4); the scratch-free
// kernel with num_levels 2-5 at compile time, the slot kernel (prefixes through HBM) with num_levels <= 5 at run time
#include "grad_wave_ho_kernel.hpp"

namespace gpsig {
typedef hipError_t (*WaveHoLaunchFn)(const WaveHoArgs&, int, size_t, hipStream_t);

#ifndef GPSIG_HO_UNDO_ONLY
template <int G, int C, int O>
static hipError_t wave_ho_launch(const WaveHoArgs& a, int nblocks, size_t, hipStream_t s) {
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
#endif

#ifndef GPSIG_HO_UNDO_ONLY
// first order from a dM lattice (seq_grad_wave_o1_kernel): 16 lanes per lattice, 2 / 4 columns per lane (lattices of at most 64 columns), 3 or 7 levels kept
template <int G, int C, int LQ>
static hipError_t wave_o1_launch(const WaveHoArgs& a, int nblocks, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((seq_grad_wave_o1_kernel<G, C, LQ>), dim3(nblocks), dim3(64), lds, s, a);
    return hipGetLastError();
}
WaveHoLaunchFn wave_o1_lookup(int G, int C, int M) {
    if (M < 1 || M > 8 || G != 16) return nullptr;
    if (C == 2) return M <= 4 ? &wave_o1_launch<16, 2, 3> : &wave_o1_launch<16, 2, 7>;
    if (C == 4) return M <= 4 ? &wave_o1_launch<16, 4, 3> : &wave_o1_launch<16, 4, 7>;
    return nullptr;
}
#endif

#ifdef GPSIG_HO_UNDO_G
template <int G, int C, int MM, int O>
static hipError_t wave_ho_undo_launch(const WaveHoArgs& a, int nblocks, size_t lds, hipStream_t s) {
    auto kern = seq_grad_wave_ho_undo_kernel<G, C, MM, O>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64), lds, s, a);
    return hipGetLastError();
}
template <int G, int C>
static WaveHoLaunchFn pick_undo(int order, int M) {
    if (M == 2 && order == 2) return &wave_ho_undo_launch<G, C, 2, 2>;
    if (M == 3 && order == 2) return &wave_ho_undo_launch<G, C, 3, 2>;
    if (M == 3 && order == 3) return &wave_ho_undo_launch<G, C, 3, 3>;
    if (M == 4 && order == 2) return &wave_ho_undo_launch<G, C, 4, 2>;
    if (M == 4 && order == 3) return &wave_ho_undo_launch<G, C, 4, 3>;
    if (M == 4 && order == 4) return &wave_ho_undo_launch<G, C, 4, 4>;
    if (M == 5 && order == 2) return &wave_ho_undo_launch<G, C, 5, 2>;
    if (M == 5 && order == 3) return &wave_ho_undo_launch<G, C, 5, 3>;
    if (M == 5 && order == 4) return &wave_ho_undo_launch<G, C, 5, 4>;
    return nullptr;
}
template <int G, int C, int MM, int O>
static hipError_t wave_ho_levels_launch(const WaveHoArgs& a, int nblocks, size_t, hipStream_t s) {
    hipLaunchKernelGGL((seq_levels_wave_ho_kernel<G, C, MM, O>), dim3(nblocks), dim3(64), 0, s, a);
    return hipGetLastError();
}
template <int G, int C>
static WaveHoLaunchFn pick_levels(int order, int M) {
    if (M == 2 && order == 2) return &wave_ho_levels_launch<G, C, 2, 2>;
    if (M == 3 && order == 2) return &wave_ho_levels_launch<G, C, 3, 2>;
    if (M == 3 && order == 3) return &wave_ho_levels_launch<G, C, 3, 3>;
    if (M == 4 && order == 2) return &wave_ho_levels_launch<G, C, 4, 2>;
    if (M == 4 && order == 3) return &wave_ho_levels_launch<G, C, 4, 3>;
    if (M == 4 && order == 4) return &wave_ho_levels_launch<G, C, 4, 4>;
    if (M == 5 && order == 2) return &wave_ho_levels_launch<G, C, 5, 2>;
    if (M == 5 && order == 3) return &wave_ho_levels_launch<G, C, 5, 3>;
    if (M == 5 && order == 4) return &wave_ho_levels_launch<G, C, 5, 4>;
    return nullptr;
}
#define HO_CAT2(a, b) a##b
#define HO_CAT(a, b) HO_CAT2(a, b)
WaveHoLaunchFn HO_CAT(wave_ho_undo_lookup_g, GPSIG_HO_UNDO_G)(int C, int order, int M) {
#if GPSIG_HO_UNDO_G == 16
    if (C == 2) return pick_undo<16, 2>(order, M);
    if (C == 4) return pick_undo<16, 4>(order, M);
#elif GPSIG_HO_UNDO_G == 32
    if (C == 2) return pick_undo<32, 2>(order, M);
#else
    if (C == 2) return pick_undo<64, 2>(order, M);
    if (C == 4) return pick_undo<64, 4>(order, M);
    if (C == 8) return pick_undo<64, 8>(order, M);
#endif
    return nullptr;
}
WaveHoLaunchFn HO_CAT(wave_ho_levels_lookup_g, GPSIG_HO_UNDO_G)(int C, int order, int M) {
#if GPSIG_HO_UNDO_G == 16
    if (C == 2) return pick_levels<16, 2>(order, M);
    if (C == 4) return pick_levels<16, 4>(order, M);
#elif GPSIG_HO_UNDO_G == 32
    if (C == 2) return pick_levels<32, 2>(order, M);
#else
    if (C == 2) return pick_levels<64, 2>(order, M);
    if (C == 4) return pick_levels<64, 4>(order, M);
    if (C == 8) return pick_levels<64, 8>(order, M);
#endif
    return nullptr;
}
#endif
}  // namespace gpsig
