// Lookup of the feature kernels (instantiated in sig_feat_inst_{a..f}.hip: D <= 32 columns, M levels with D^M <= SIG_MAX_TOP) and the
// launchers of the contraction and its reduce.
#include "sig_feat_kernel.hpp"

namespace gpsig {
typedef hipError_t (*SigFeatLaunchFn)(const SigFeatArgs&, unsigned, size_t, hipStream_t);
SigFeatLaunchFn sig_feat_pick_a(int d, int M);       // d = 1 .. 4     (sig_feat_inst_a.hip)
SigFeatLaunchFn sig_feat_pick_b(int d, int M);       // d = 5 .. 8
SigFeatLaunchFn sig_feat_pick_c(int d, int M);       // d = 9 .. 12
SigFeatLaunchFn sig_feat_pick_d(int d, int M);       // d = 13 .. 16
SigFeatLaunchFn sig_feat_pick_e(int d, int M);       // d = 17 .. 24
SigFeatLaunchFn sig_feat_pick_f(int d, int M);       // d = 25 .. 32

SigFeatLaunchFn sig_feat_lookup(int d, int M) {
    if (d < 1 || d > 32) return nullptr;
    return d <= 4 ? sig_feat_pick_a(d, M) : d <= 8 ? sig_feat_pick_b(d, M) : d <= 12 ? sig_feat_pick_c(d, M) : d <= 16 ? sig_feat_pick_d(d, M)
         : d <= 24 ? sig_feat_pick_e(d, M) : sig_feat_pick_f(d, M);
}

hipError_t sig_gram_launch(const SigGramArgs& G, int ntiles, hipStream_t stream, int dma, int* used_dma) {
    if (used_dma) *used_dma = 0;
    const bool aligned = (G.lda % 2) == 0 && (G.ldb % 2) == 0 && (reinterpret_cast<uintptr_t>(G.A) % 16) == 0 &&
                         (reinterpret_cast<uintptr_t>(G.B) % 16) == 0;
    const int ke_pad = (G.k_end + SG_BK - 1) / SG_BK * SG_BK;
    if (dma && aligned && G.k_begin % SG_BK == 0 && ke_pad <= G.lda && ke_pad <= G.ldb) {     // whole slabs, zeros behind k_end
        hipLaunchKernelGGL(sig_gram_dma_kernel, dim3(unsigned(ntiles) * unsigned(G.nsplit)), dim3(256), 0, stream, G);
        if (used_dma) *used_dma = 1;
        return hipGetLastError();
    }
    const bool vec = (G.k_begin % 2) == 0 && (G.lda % 2) == 0 && (G.ldb % 2) == 0 && (reinterpret_cast<uintptr_t>(G.A) % 16) == 0 &&
                     (reinterpret_cast<uintptr_t>(G.B) % 16) == 0;
    if (vec) hipLaunchKernelGGL(sig_gram_kernel<true>, dim3(unsigned(ntiles) * unsigned(G.nsplit)), dim3(256), 0, stream, G);
    else hipLaunchKernelGGL(sig_gram_kernel<false>, dim3(unsigned(ntiles) * unsigned(G.nsplit)), dim3(256), 0, stream, G);
    return hipGetLastError();
}
hipError_t sig_convert_launch(const void* in, void* out, int64_t n, bool widen, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t g = (n + 255) / 256;
    if (g > 16384) g = 16384;
    if (widen) hipLaunchKernelGGL(sig_widen_kernel, dim3(unsigned(g)), dim3(256), 0, stream, static_cast<const float*>(in), static_cast<double*>(out), n);
    else hipLaunchKernelGGL(sig_narrow_kernel, dim3(unsigned(g)), dim3(256), 0, stream, static_cast<const double*>(in), static_cast<float*>(out), n);
    return hipGetLastError();
}
hipError_t sig_reduce_launch(const SigReduceArgs& R, hipStream_t stream) {
    if (R.mode == 1) {                   // symmetric: per computed tile, mirrored through LDS
        const int nt = int((R.NA + SG_BM - 1) / SG_BM);
        hipLaunchKernelGGL(sig_gram_reduce_sym_kernel, dim3((SG_BM / 32) * (SG_BM / 32), unsigned(nt * (nt + 1) / 2)), dim3(32, 8), 0, stream, R, nt);
        return hipGetLastError();
    }
    int64_t g = (R.NA * R.NB + 255) / 256;
    if (g > 65536) g = 65536;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(sig_gram_reduce_kernel, dim3(unsigned(g)), dim3(256), 0, stream, R);
    return hipGetLastError();
}
}  // namespace gpsig
