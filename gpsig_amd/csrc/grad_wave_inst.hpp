// grad_wave_inst.hpp -- instantiation list of seq_grad_wave_kernel for one lattice mode (included by grad_wave_inst_*.hip)
#pragma once
#include "grad_wave_kernel.hpp"

namespace gpsig {
typedef hipError_t (*WaveLaunchFn)(const WaveGradArgs&, int, hipStream_t);

template <int G, int C, int DP, int LQ, int MODE>
hipError_t wave_launch(const WaveGradArgs& a, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL((seq_grad_wave_kernel<G, C, DP, LQ, MODE>), dim3(nblocks), dim3(64), 0, s, a);
    return hipGetLastError();
}

// (G, C, DP): lanes per pair, columns per lane, padded feature count.  C * DP bounded by the register file.
#define GPSIG_WAVE_SHAPES(X) \
    X(16, 2, 4) X(16, 2, 8) X(16, 2, 16) X(16, 4, 4) X(16, 4, 8) X(16, 4, 16) \
    X(64, 2, 4) X(64, 2, 8) X(64, 2, 16) X(64, 4, 4) X(64, 4, 8) X(64, 4, 16) X(64, 8, 4) X(64, 8, 8)

typedef hipError_t (*Wave2LaunchFn)(const Wave2Args&, int, size_t, hipStream_t);
// Level variants of the scratch-free kernels: num_levels 4 and 5 at compile time (LQ = 3, 4), otherwise LQ = 4 or 7 at run time.
#define GPSIG_W2_PICK(LAUNCH, G_, C_, D_)                                           \
    if (G == G_ && C == C_ && DP == D_)                                             \
        return LQ == 3 ? LAUNCH<G_, C_, D_, 3, MODE, true>                          \
                       : (LQ == 4 ? LAUNCH<G_, C_, D_, 4, MODE, true> : (LQ < 3 ? LAUNCH<G_, C_, D_, 4, MODE, false> : LAUNCH<G_, C_, D_, 7, MODE, false>));

#ifdef GPSIG_INST_WAVE2
template <int G, int C, int DP, int LQ, int MODE, bool MX>
hipError_t wave2_launch(const Wave2Args& a, int nblocks, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((seq_grad_wave2_kernel<G, C, DP, LQ, MODE, MX>), dim3(nblocks), dim3(64), lds, s, a);
    return hipGetLastError();
}
template <int MODE>
Wave2LaunchFn wave2_lookup_mode(int G, int C, int DP, int LQ) {
#define X_W2(G_, C_, D_) GPSIG_W2_PICK(wave2_launch, G_, C_, D_)
    GPSIG_WAVE_SHAPES(X_W2)
#undef X_W2
    return nullptr;
}
#endif

#ifdef GPSIG_INST_LAM
// KIND is part of every name here: the translation units of the two kernel families instantiate different functions
template <int KIND>
struct LamUndoInst {
    template <int G, int C, int DP, int LQ, int MODE, bool MX>
    static hipError_t launch(const Wave2Args& a, int nblocks, size_t lds, hipStream_t s) {
        hipLaunchKernelGGL((seq_lam_undo_kernel<G, C, DP, LQ, MODE, MX, KIND>), dim3(nblocks), dim3(64), lds, s, a);
        return hipGetLastError();
    }
    template <int MODE>
    static Wave2LaunchFn lookup(int G, int C, int DP, int LQ) {
#define X_LU(G_, C_, D_) GPSIG_W2_PICK(launch, G_, C_, D_)
        GPSIG_WAVE_SHAPES(X_LU)
#undef X_LU
        return nullptr;
    }
};
#endif

template <int MODE>
WaveLaunchFn wave_lookup_mode(int G, int C, int DP, int LQ) {
#define X_W(G_, C_, D_)                                                              \
    if (G == G_ && C == C_ && DP == D_) return LQ <= 4 ? wave_launch<G_, C_, D_, 4, MODE> : wave_launch<G_, C_, D_, 7, MODE>;
    GPSIG_WAVE_SHAPES(X_W)
#undef X_W
    return nullptr;
}
}  // namespace gpsig
