// Instantiations of sig_features_kernel (sig_feat_kernel.hpp) and their lookup: D columns, M levels with D^M <= SIG_MAX_TOP.
#include "sig_feat_kernel.hpp"

namespace gpsig {
typedef hipError_t (*SigFeatLaunchFn)(const SigFeatArgs&, unsigned, size_t, hipStream_t);

template <int D, int M>
static hipError_t sig_feat_launch(const SigFeatArgs& A, unsigned grid, size_t lds, hipStream_t stream) {
    auto kern = sig_features_kernel<D, M, false>;
    if constexpr (sig_siblings(D, M)) kern = sig_features_sib_kernel<D, M, false>;      // sibling parents per thread: fewer multiply-adds
    if (A.order > 1) {                               // the higher-order algorithm: truncated-exponential steps
        kern = sig_features_kernel<D, M, true>;
        if constexpr (sig_siblings(D, M)) kern = sig_features_sib_kernel<D, M, true>;
    }
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(sig_threads(D, M)), lds, stream, A);
    return hipGetLastError();
}

template <int D>
static SigFeatLaunchFn sig_feat_pick(int M) {
    switch (M) {
        case 2: return &sig_feat_launch<D, 2>;
        case 3: return &sig_feat_launch<D, 3>;
        case 4: if constexpr (sig_ipow(D, 4) <= SIG_MAX_TOP) return &sig_feat_launch<D, 4>; else return nullptr;
        case 5: if constexpr (sig_ipow(D, 5) <= SIG_MAX_TOP) return &sig_feat_launch<D, 5>; else return nullptr;
        case 6: if constexpr (sig_ipow(D, 5) <= SIG_MAX_TOP && sig_ipow(D, 6) <= SIG_MAX_TOP) return &sig_feat_launch<D, 6>; else return nullptr;
        case 7: if constexpr (sig_ipow(D, 6) <= SIG_MAX_TOP && sig_ipow(D, 7) <= SIG_MAX_TOP) return &sig_feat_launch<D, 7>; else return nullptr;
        case 8: if constexpr (sig_ipow(D, 7) <= SIG_MAX_TOP && sig_ipow(D, 8) <= SIG_MAX_TOP) return &sig_feat_launch<D, 8>; else return nullptr;
        default: return nullptr;
    }
}

SigFeatLaunchFn sig_feat_lookup(int d, int M) {
    switch (d) {
        case 1: return sig_feat_pick<1>(M);
        case 2: return sig_feat_pick<2>(M);
        case 3: return sig_feat_pick<3>(M);
        case 4: return sig_feat_pick<4>(M);
        case 5: return sig_feat_pick<5>(M);
        case 6: return sig_feat_pick<6>(M);
        case 7: return sig_feat_pick<7>(M);
        case 8: return sig_feat_pick<8>(M);
        default: return nullptr;
    }
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
