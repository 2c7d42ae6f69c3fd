// tensor-vs-sequence kernel instantiations, tensor-lane variant (one lane per inducing tensor): see tens_inst.hip
#include "aux_kernels.hpp"

namespace gpsig {
// ---- tensor-lane variant: levels are swept in groups [LO, HI] whose components fit the register file
typedef hipError_t (*TvsLaneTLaunchFn)(const TvsLaneTArgs&, hipStream_t);

template <int LO, int HI, int D, bool INCR>
static hipError_t tvs_lanet_launch(const TvsLaneTArgs& A, hipStream_t stream) {
    dim3 grid((unsigned)(A.Tpad / 64), (unsigned)A.N);
    const size_t lds = sizeof(double) * size_t(A.L) * A.d_eff;
    hipLaunchKernelGGL((tens_vs_seq_lanet_kernel<double, LO, HI, D, INCR>), grid, dim3(64), lds, stream, A);
    return hipGetLastError();
}

// groups available in this build
#define TVL_GROUPS(X) X(1, 1) X(1, 2) X(1, 3) X(1, 4) X(1, 5) X(2, 2) X(3, 3) X(4, 4) X(5, 5) X(6, 6) X(4, 5) X(3, 4)

template <int D, bool INCR>
static TvsLaneTLaunchFn tvl_group(int lo, int hi) {
#define TVL_CASE(LO_, HI_) if (lo == LO_ && hi == HI_) return &tvs_lanet_launch<LO_, HI_, D, INCR>;
    TVL_GROUPS(TVL_CASE)
#undef TVL_CASE
    return nullptr;
}

static TvsLaneTLaunchFn tvl_lookup(int lo, int hi, int D, bool incr) {
    if (D == 4) return incr ? tvl_group<4, true>(lo, hi) : tvl_group<4, false>(lo, hi);
    if (D == 8) return incr ? tvl_group<8, true>(lo, hi) : tvl_group<8, false>(lo, hi);
    return incr ? tvl_group<16, true>(lo, hi) : tvl_group<16, false>(lo, hi);
}

// Split levels 1..M into consecutive groups of at most `budget` doubles of components per lane.  Fills fns/ngroups;
// returns false if some level alone does not fit or a needed group is not built (caller falls back).
bool tvs_lanet_plan(int M, int d, bool incr, TvsLaneTLaunchFn* fns, int* ngroups) {
    if (d > 16 || M > 6 || M < 1) return false;
    const int D = d <= 4 ? 4 : (d <= 8 ? 8 : 16);
    const int per_comp = D * (incr ? 2 : 1), budget = 64;   // doubles of components per lane: 2 waves per SIMD fit
    int n = 0, lo = 1;
    while (lo <= M) {
        int hi = lo, comps = lo;
        if (comps * per_comp > budget) return false;
        while (hi + 1 <= M && (comps + hi + 1) * per_comp <= budget) { ++hi; comps += hi; }
        TvsLaneTLaunchFn f = tvl_lookup(lo, hi, D, incr);
        while (!f && hi > lo) { --hi; f = tvl_lookup(lo, hi, D, incr); }     // fall back to a smaller built group
        if (!f) return false;
        fns[n++] = f;
        lo = hi + 1;
    }
    *ngroups = n;
    return true;
}
}  // namespace gpsig
