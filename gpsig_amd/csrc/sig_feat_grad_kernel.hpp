// sig_feat_grad_kernel.hpp -- reverse pass of the explicit level features (sig_feat_kernel.hpp): the gradient of SignatureLinear's
// sequence-vs-sequence levels through the feature contraction (round 4).
//
// The reference trains through TensorFlow's autodiff of signature_algs.py:8-35; for the LINEAR state-space kernel level m of a pair is
// K_m(x, y) = < Phi_m(x), Phi_m(y) > with Phi_m built by the sweep  Phi_m <- Phi_m + Phi_{m-1} (x) dx_a  over the increments (header of
// sig_feat_kernel.hpp).  Given the upstream G (M+1, N1, N2) of the level array, the gradient with respect to the features is a plain
// product per level, dPhi_m(X) = G_m Phi_m(Y) (rocBLAS dgemm, api side), and this kernel takes dPhi back through the sweep to the
// sequence:  one workgroup per sequence walks the time steps BACKWARDS with, per level n < M, the features phi_n of the state BEFORE the
// step (the forward sweep is undone, Phi_n^{a-1} = Phi_n^a - Phi_{n-1}^{a-1} (x) dx_a: nothing of the forward pass is stored but its
// final value) and the adjoints lam_n = dL/dPhi_n after the step:
//     g_a[f]            = sum_{n < M} sum_I phi_n^{a-1}[I] lam_{n+1}^a[I, f]            (the gradient with respect to the increment dx_a)
//     lam_n^{a-1}[I]    = lam_n^a[I] + sum_f lam_{n+1}^a[I, f] dx_a[f]                   (lam_M is the upstream itself, constant)
// Layout of the work: "item" (n, I) -- a multi-index I of length n -- owns phi_n[I], lam_n[I] and row I of level n+1's adjoint.  The
// rows of the TOP level (n = M-1: d^(M-1) rows of d upstream values, nearly all of the arithmetic) live in registers, cyclically over
// the threads like the parents of sig_features_kernel; every lower level lives in LDS (two copies: a step reads the values of time a
// and writes those of time a-1).  Two barriers per step: the low items first (their undo needs the chain of ancestors, a few entries),
// then the top rows, which read their parent's new value.  The per-thread partial sums of g_a are added up per wavefront through a
// wave-private LDS tile and land in per-wave slots (a fixed order of summation: deterministic).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sig_feat_kernel.hpp"

namespace gpsig {

struct SigFeatGradArgs {
    const double* X;        // (N, L, D) sequences as given (the level primitives take their inputs unscaled)
    int64_t N;
    int L, difference;
    const double* Phi;      // (N, ld) features in natural order (sig_features_kernel, no weights, no normalisation): levels 1..M, level 0
    const double* dPhi;     // (N, ld) upstream gradient with respect to them, same layout
    int64_t ld;
    double* gX;             // (N, L, D) out
    // higher orders (signature_algs.py:37-74; sig_feat_reverse_ho_kernel): order, the inverse series of the truncated exponential
    // (cinv[k]: Phi^{a-1} = Phi^a (x) sum_k cinv[k] dx^(x)k) and the weights of the Horner sub-steps' intermediates
    // (T^(j) = sum_k w[j][k] Phi^{a-1} (x) dx^(x)k, w[j][k] = j! / (j+k)!)
    int order;
    double cinv[9];
    double w[9][9];
    int unit_points;        // SignatureCosine: the features are those of the unit vectors x / |x| (SigFeatArgs::unit_points); the gradient is
                            // taken on through the normalisation, d(x / |x|) = (I - u u^T) / |x|
};

constexpr int sig_geo(int D, int n) { int s = 0; for (int k = 0; k < n; ++k) s += sig_ipow(D, k); return s; }      // 1 + D + .. + D^(n-1)
constexpr int sig_grad_low_count(int D, int M) { return sig_geo(D, M - 1); }                                         // items of levels 0 .. M-2
constexpr int SIG_GRAD_TCH = 8;             // time steps whose per-wave partial sums are held before they are added up

// dynamic LDS: increments (R x D), the sum g (R x D), phi of levels 0 .. M-2 and lam of levels 1 .. M-1 twice, wave tiles, per-wave slots
inline size_t sig_feat_grad_lds_bytes(int D, int M, int L) {
    const int T = sig_threads(D, M), W = T / 64, FB = D < 8 ? D : 8;
    const size_t lam = size_t(sig_geo(D, M)) - 1;
    const size_t doubles = 2 * size_t(L) * D + 2 * size_t(sig_grad_low_count(D, M)) + 2 * lam + size_t(W) * FB * 64 + size_t(SIG_GRAD_TCH) * W * D + 16;
    return sizeof(double) * doubles;
}

// the higher-order kernel: lam of levels 1 .. M-2 twice, level M-1 once, the sub-steps' adjoints tau (levels 1 .. M-1 and 1 .. M-2)
inline size_t sig_feat_grad_ho_lds_bytes(int D, int M, int L) {
    const int T = sig_threads(D, M), W = T / 64, FB = D < 8 ? D : 8;
    const size_t lamtot = size_t(sig_geo(D, M)) - 1, lamlow = size_t(sig_geo(D, M - 1)) - 1 + 1, ntop = size_t(sig_ipow(D, M - 1));
    const size_t doubles = 2 * size_t(L) * D + 2 * size_t(sig_grad_low_count(D, M)) + 2 * lamlow + ntop + lamtot + lamlow + size_t(W) * FB * 64 +
                           size_t(SIG_GRAD_TCH) * W * D + 16;
    return sizeof(double) * doubles;
}

template <int D, int M>
__global__ __launch_bounds__(sig_threads(D, M)) void sig_feat_reverse_kernel(const SigFeatGradArgs A) {
    static_assert(M >= 2, "levels");
    constexpr int T = sig_threads(D, M), W = T / 64;
    constexpr int NTOPROWS = sig_ipow(D, M - 1);                        // rows of the top level's adjoint = entries of level M-1
    constexpr int PPT = (NTOPROWS + T - 1) / T;
    constexpr int NLOW = sig_grad_low_count(D, M);                      // items (n, I), n = 0 .. M-2
    constexpr int LPT = (NLOW + T - 1) / T;                             // low items per thread
    constexpr int FB = D < 8 ? D : 8;                                   // components reduced per pass of the wave tile
    constexpr int GL = (64 / FB) >= 32 ? 32 : (64 / FB) >= 16 ? 16 : (64 / FB) >= 8 ? 8 : (64 / FB) >= 4 ? 4 : 2;    // lanes per component
    extern __shared__ double sg_sm[];
    const int R = A.difference ? A.L - 1 : A.L;
    // level n starts at sig_geo(D, n) in the phi array (levels 0 .. M-2), at sig_geo(D, n) - 1 in the lam array (levels 1 .. M-1) and in a
    // feature row (levels 1 .. M, natural order)
    constexpr int lamtot = sig_geo(D, M) - 1;
    double* const dx = sg_sm;                                   // R x D
    double* const gsum = dx + size_t(A.L) * D;                  // R x D
    double* const phiB = gsum + size_t(A.L) * D;                // 2 x NLOW
    double* const lamB = phiB + 2 * NLOW;                       // 2 x lamtot
    double* const tile = lamB + 2 * size_t(lamtot);             // W x FB x 64
    double* const slots = tile + size_t(W) * FB * 64;           // TCH x W x D
    auto phi_off = [](int n) { return sig_geo(D, n); };
    auto lam_off = [](int n) { return sig_geo(D, n) - 1; };
    auto feat_off = [](int m) { return sig_geo(D, m) - 1; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the thread's top rows I = q T + tid: parent entry (level M-2) and last component
    // (where D divides T both are affine in q -- one register each instead of one per row)
    constexpr bool AFFINE = T % D == 0;
    int tparent_[AFFINE ? 1 : PPT], tcomp_[AFFINE ? 1 : PPT];
    if constexpr (AFFINE) {
        tparent_[0] = phi_off(M - 2) + tid / D;
        tcomp_[0] = tid % D;
    } else {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int I = q * T + tid;
            tparent_[q] = phi_off(M - 2) + (I < NTOPROWS ? I / D : 0);
            tcomp_[q] = I % D;
        }
    }
    auto tparent = [&](int q) {
        if constexpr (AFFINE) { const int v = tparent_[0] + q * (T / D); return v < phi_off(M - 2) + sig_ipow(D, M - 2) ? v : phi_off(M - 2); }
        else return tparent_[q];
    };
    auto tcomp = [&](int q) { if constexpr (AFFINE) return tcomp_[0]; else return tcomp_[q]; };
    // the thread's low items: level (-1: none), where the item's own phi / lam sit, where its row of the level above starts, and the chain
    // of its ancestors (prefix of length j: its place in phi and the component of dx that extends the prefix of length j-1 to it)
    constexpr int NCH = M >= 3 ? M - 2 : 1;
    int ln[LPT], lphi[LPT], llam[LPT], lrow[LPT], choff[LPT][NCH], chcmp[LPT][NCH];
#pragma unroll
    for (int s = 0; s < LPT; ++s) {
        int it = s * T + tid, n = -1, I = 0;
        if (it < NLOW) {
            n = 0;
            while (it >= sig_ipow(D, n)) { it -= sig_ipow(D, n); ++n; }
            I = it;
        }
        ln[s] = n;
        lphi[s] = (n >= 0 ? phi_off(n) : 0) + I;
        llam[s] = (n >= 1 ? lam_off(n) : 0) + I;
        lrow[s] = (n >= 0 ? lam_off(n + 1) : 0) + I * D;
#pragma unroll
        for (int j = 1; j <= NCH; ++j) {
            int pre = I;
            for (int k = j; k < n; ++k) pre /= D;                   // prefix of length j of an index of length n
            choff[s][j - 1] = (j <= n ? phi_off(j) : 0) + (j <= n ? pre : 0);
            chcmp[s][j - 1] = j <= n ? pre % D : 0;
        }
    }
    for (int64_t sq = blockIdx.x; sq < A.N; sq += gridDim.x) {
        const double* Xn = A.X + sq * int64_t(A.L) * D;
        const double* Ph = A.Phi + sq * A.ld;
        const double* dP = A.dPhi + sq * A.ld;
        __syncthreads();
        if (!A.unit_points) {
            for (int e = tid; e < R * D; e += T) {
                const int a = e / D, f = e - a * D;
                dx[e] = A.difference ? Xn[(a + 1) * D + f] - Xn[a * D + f] : Xn[e];
            }
        } else {                                    // the unit vectors first (in the space of the sum g, which is not written before the sweep)
            for (int t = tid; t < A.L; t += T) {
                double ss = 0.0;
#pragma unroll
                for (int f = 0; f < D; ++f) ss = fma(Xn[t * D + f], Xn[t * D + f], ss);
                const double inv = 1.0 / sqrt(ss);
#pragma unroll
                for (int f = 0; f < D; ++f) gsum[t * D + f] = Xn[t * D + f] * inv;
            }
            __syncthreads();
            for (int e = tid; e < R * D; e += T) {
                const int a = e / D, f = e - a * D;
                dx[e] = A.difference ? gsum[(a + 1) * D + f] - gsum[a * D + f] : gsum[e];
            }
        }
        // state after the last step: the forward pass's final features, the upstream gradients
        for (int e = tid; e < NLOW; e += T) phiB[e] = e == 0 ? 1.0 : Ph[e - 1];            // level 0 == 1; levels 1 .. M-2 follow in natural order
        for (int e = tid; e < lamtot; e += T) lamB[e] = dP[e];                               // levels 1 .. M-1
        double Gtop[PPT][D], phiTop[PPT], lamTop[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int I = q * T + tid;
            const bool ok = I < NTOPROWS;
            phiTop[q] = ok ? Ph[feat_off(M - 1) + I] : 0.0;
            lamTop[q] = ok ? dP[feat_off(M - 1) + I] : 0.0;
#pragma unroll
            for (int f = 0; f < D; ++f) Gtop[q][f] = ok ? dP[feat_off(M) + int64_t(I) * D + f] : 0.0;
        }
        __syncthreads();
        int cur = 0;
        for (int a = R - 1; a >= 0; --a) {
            const double* const dxa = dx + a * D;
            const double* const phiC = phiB + cur * NLOW;
            double* const phiN = phiB + (cur ^ 1) * NLOW;
            const double* const lamC = lamB + cur * lamtot;
            double* const lamN = lamB + (cur ^ 1) * lamtot;
            double d_[D], gp[D];
#pragma unroll
            for (int f = 0; f < D; ++f) { d_[f] = dxa[f]; gp[f] = 0.0; }
            // ---- low items: undo phi along the chain of ancestors (old values only), partial g, lam one level down
#pragma unroll
            for (int s = 0; s < LPT; ++s) {
                const int n = ln[s];
                if (n < 0) continue;
                // phi_n^{a-1}[I] = h_n,  h_0 = 1,  h_j = phi_j^a[I_{1..j}] - h_{j-1} dx[i_j]
                double h = 1.0;
#pragma unroll
                for (int j = 1; j <= NCH; ++j)
                    if (j <= n) h = phiC[choff[s][j - 1]] - h * dxa[chcmp[s][j - 1]];
                phiN[lphi[s]] = h;                                          // (level 0 stays 1)
                // row I of level n+1's adjoint (time a): children I D + f
                const double* row = lamC + lrow[s];
                double acc = 0.0;
#pragma unroll
                for (int f = 0; f < D; ++f) {
                    const double l = row[f];
                    gp[f] = fma(h, l, gp[f]);
                    acc = fma(l, d_[f], acc);
                }
                if (n >= 1) lamN[llam[s]] = lamC[llam[s]] + acc;
            }
            __syncthreads();                                                // the new phi of level M-2 is complete
            // ---- top rows: registers
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const double hp = phiN[tparent(q)];                          // phi_{M-2}^{a-1}[parent]  (M == 2: level 0 == 1)
                const double ph = phiTop[q] - hp * dxa[tcomp(q)];            // phi_{M-1}^{a-1}[I]
                phiTop[q] = ph;
                double acc = 0.0;
#pragma unroll
                for (int f = 0; f < D; ++f) {
                    gp[f] = fma(ph, Gtop[q][f], gp[f]);
                    acc = fma(Gtop[q][f], d_[f], acc);
                }
                const double lnew = lamTop[q] + acc;
                // the row's own adjoint at time a is what the level below read in this step (lamC); publish the new one
                const int I = q * T + tid;
                if (I < NTOPROWS) lamN[lam_off(M - 1) + I] = lnew;
                lamTop[q] = lnew;
            }
            // ---- g_a: per-thread partial sums -> per-wave sums (wave-private tile, FB components per pass) -> slot of this step
            double* const wt = tile + size_t(wave) * FB * 64;
            double* const slot = slots + (size_t(a % SIG_GRAD_TCH) * W + wave) * D;
#pragma unroll
            for (int f0 = 0; f0 < D; f0 += FB) {
#pragma unroll
                for (int f = 0; f < FB; ++f)
                    if (f0 + f < D) wt[f * 64 + lane] = gp[f0 + f];
                // (same wavefront: the LDS unit executes a wave's accesses in order)
                const int fl = lane / GL, sub = lane % GL;
                double s = 0.0;
                if (fl < FB && f0 + fl < D) {
#pragma unroll
                    for (int k = 0; k < 64 / GL; ++k) s += wt[fl * 64 + sub + GL * k];
                }
#pragma unroll
                for (int o = GL / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                if (fl < FB && f0 + fl < D && sub == 0) slot[f0 + fl] = s;
            }
            __syncthreads();                                                // lamN / phiN complete; slots of this step written
            if (a % SIG_GRAD_TCH == 0 || a == 0) {                          // add up the waves' sums of the steps a .. (held ones)
                const int a_hi = (a / SIG_GRAD_TCH) * SIG_GRAD_TCH + SIG_GRAD_TCH - 1 < R - 1 ? (a / SIG_GRAD_TCH) * SIG_GRAD_TCH + SIG_GRAD_TCH - 1 : R - 1;
                for (int e = tid; e < (a_hi - a + 1) * D; e += T) {
                    const int aa = a + e / D, f = e % D;
                    double s = 0.0;
                    for (int w = 0; w < W; ++w) s += slots[(size_t(aa % SIG_GRAD_TCH) * W + w) * D + f];
                    gsum[aa * D + f] = s;
                }
                // (the next step's slot writes come after its first barrier: no race with these reads)
            }
            cur ^= 1;
        }
        __syncthreads();
        // d/dX from d/d(increments): signature_algs.py:25-26 takes differences of the kernel matrix = increments of the sequence here
        double* gx = A.gX + sq * int64_t(A.L) * D;
        if (!A.unit_points) {
            for (int e = tid; e < A.L * D; e += T) {
                const int t = e / D, f = e - t * D;
                double v;
                if (A.difference) v = (t >= 1 ? gsum[(t - 1) * D + f] : 0.0) - (t < R ? gsum[t * D + f] : 0.0);
                else v = gsum[e];
                gx[e] = v;
            }
        } else {                                    // through u = x / |x|:  gx = (gu - u <u, gu>) / |x|
            for (int t = tid; t < A.L; t += T) {
                double gu[D], u[D], ss = 0.0, dot = 0.0;
#pragma unroll
                for (int f = 0; f < D; ++f) {
                    gu[f] = A.difference ? (t >= 1 ? gsum[(t - 1) * D + f] : 0.0) - (t < R ? gsum[t * D + f] : 0.0) : gsum[t * D + f];
                    u[f] = Xn[t * D + f];
                    ss = fma(u[f], u[f], ss);
                }
                const double inv = 1.0 / sqrt(ss);
#pragma unroll
                for (int f = 0; f < D; ++f) { u[f] *= inv; dot = fma(u[f], gu[f], dot); }
#pragma unroll
                for (int f = 0; f < D; ++f) gx[t * D + f] = (gu[f] - u[f] * dot) * inv;
            }
        }
    }
}

// The same reverse pass for the higher-order algorithm (signature_algs.py:37-74 with the linear kernel: a step multiplies by the exponential of the
// increment truncated at degree `order`, sig_horner in sig_feat_kernel.hpp).  Undoing a step is multiplying by the inverse series; the step's adjoint
// is taken through its `order` Horner sub-steps, one phase (and barrier) each -- but only the levels below the top take part beyond phase 0: the top
// level's rows, nearly all of the arithmetic, do what they do at order 1.
template <int D, int M>
__global__ __launch_bounds__(sig_threads(D, M)) void sig_feat_reverse_ho_kernel(const SigFeatGradArgs A) {
    static_assert(M >= 2, "levels");
    constexpr int T = sig_threads(D, M), W = T / 64;
    constexpr int NTOPROWS = sig_ipow(D, M - 1);                        // rows of the top level's adjoint = entries of level M-1
    constexpr int PPT = (NTOPROWS + T - 1) / T;
    constexpr int NLOW = sig_grad_low_count(D, M);                      // items (n, I), n = 0 .. M-2
    constexpr int LPT = (NLOW + T - 1) / T;                             // low items per thread
    constexpr int FB = D < 8 ? D : 8;                                   // components reduced per pass of the wave tile
    constexpr int GL = (64 / FB) >= 32 ? 32 : (64 / FB) >= 16 ? 16 : (64 / FB) >= 8 ? 8 : (64 / FB) >= 4 ? 4 : 2;    // lanes per component
    extern __shared__ double sg_sm[];
    const int R = A.difference ? A.L - 1 : A.L;
    // level n starts at sig_geo(D, n) in the phi array (levels 0 .. M-2), at sig_geo(D, n) - 1 in the lam array (levels 1 .. M-1) and in a
    // feature row (levels 1 .. M, natural order)
    constexpr int lamtot = sig_geo(D, M) - 1;
    double* const dx = sg_sm;                                   // R x D
    double* const gsum = dx + size_t(A.L) * D;                  // R x D
    double* const phiB = gsum + size_t(A.L) * D;                // 2 x NLOW
    constexpr int lamlow = sig_geo(D, M - 1) - 1 + 1;           // levels 1 .. M-2 (+ 1: never empty)
    double* const lamB = phiB + 2 * NLOW;                       // 2 x lamlow: lam of levels 1 .. M-2 at time a and a-1
    double* const lamM1 = lamB + 2 * lamlow;                    // NTOPROWS: lam of level M-1 (read in phase 0, rewritten behind it)
    double* const tauA = lamM1 + NTOPROWS;                      // lamtot: the sub-steps' adjoints of odd phases (levels 1 .. M-1)
    double* const tauB = tauA + lamtot;                         // lamlow: ... of even phases (levels 1 .. M-2)
    double* const tile = tauB + lamlow;                         // W x FB x 64
    double* const slots = tile + size_t(W) * FB * 64;           // TCH x W x D
    auto phi_off = [](int n) { return sig_geo(D, n); };
    auto lam_off = [](int n) { return sig_geo(D, n) - 1; };
    auto feat_off = [](int m) { return sig_geo(D, m) - 1; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the thread's top rows I = q T + tid: the places of their prefixes of length 1 .. M-2 in phi and the components that extend a prefix
    // by one index (tcmp[q][j-1]: the last index of the prefix of length j, j = 1 .. M-1)
    constexpr int NCT = M >= 3 ? M - 2 : 1;
    int tpre[PPT][NCT], tcmp[PPT][NCT + 1];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int I = q * T + tid;
        const bool ok = I < NTOPROWS;
#pragma unroll
        for (int j = 1; j <= M - 1; ++j) {
            int pre = ok ? I : 0;
            for (int k = j; k < M - 1; ++k) pre /= D;
            if (j <= M - 2) tpre[q][j - 1] = phi_off(j) + pre;
            tcmp[q][j - 1] = pre % D;
        }
    }
    const int P = A.order;
    // the thread's low items: level (-1: none), where the item's own phi / lam sit, where its row of the level above starts, and the chain
    // of its ancestors (prefix of length j: its place in phi and the component of dx that extends the prefix of length j-1 to it)
    constexpr int NCH = M >= 3 ? M - 2 : 1;
    int ln[LPT], lphi[LPT], llam[LPT], lrow[LPT], choff[LPT][NCH], chcmp[LPT][NCH];
#pragma unroll
    for (int s = 0; s < LPT; ++s) {
        int it = s * T + tid, n = -1, I = 0;
        if (it < NLOW) {
            n = 0;
            while (it >= sig_ipow(D, n)) { it -= sig_ipow(D, n); ++n; }
            I = it;
        }
        ln[s] = n;
        lphi[s] = (n >= 0 ? phi_off(n) : 0) + I;
        llam[s] = (n >= 1 ? lam_off(n) : 0) + I;
        lrow[s] = (n >= 0 ? lam_off(n + 1) : 0) + I * D;
#pragma unroll
        for (int j = 1; j <= NCH; ++j) {
            int pre = I;
            for (int k = j; k < n; ++k) pre /= D;                   // prefix of length j of an index of length n
            choff[s][j - 1] = (j <= n ? phi_off(j) : 0) + (j <= n ? pre : 0);
            chcmp[s][j - 1] = j <= n ? pre % D : 0;
        }
    }
    for (int64_t sq = blockIdx.x; sq < A.N; sq += gridDim.x) {
        const double* Xn = A.X + sq * int64_t(A.L) * D;
        const double* Ph = A.Phi + sq * A.ld;
        const double* dP = A.dPhi + sq * A.ld;
        __syncthreads();
        if (!A.unit_points) {
            for (int e = tid; e < R * D; e += T) {
                const int a = e / D, f = e - a * D;
                dx[e] = A.difference ? Xn[(a + 1) * D + f] - Xn[a * D + f] : Xn[e];
            }
        } else {                                    // the unit vectors first (in the space of the sum g, which is not written before the sweep)
            for (int t = tid; t < A.L; t += T) {
                double ss = 0.0;
#pragma unroll
                for (int f = 0; f < D; ++f) ss = fma(Xn[t * D + f], Xn[t * D + f], ss);
                const double inv = 1.0 / sqrt(ss);
#pragma unroll
                for (int f = 0; f < D; ++f) gsum[t * D + f] = Xn[t * D + f] * inv;
            }
            __syncthreads();
            for (int e = tid; e < R * D; e += T) {
                const int a = e / D, f = e - a * D;
                dx[e] = A.difference ? gsum[(a + 1) * D + f] - gsum[a * D + f] : gsum[e];
            }
        }
        // state after the last step: the forward pass's final features, the upstream gradients
        for (int e = tid; e < NLOW; e += T) phiB[e] = e == 0 ? 1.0 : Ph[e - 1];            // level 0 == 1; levels 1 .. M-2 follow in natural order
        for (int e = tid; e < lamlow - 1; e += T) lamB[e] = dP[e];                           // levels 1 .. M-2
        for (int e = tid; e < NTOPROWS; e += T) lamM1[e] = dP[feat_off(M - 1) + e];          // level M-1
        double Gtop[PPT][D], phiTop[PPT], lamTop[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int I = q * T + tid;
            const bool ok = I < NTOPROWS;
            phiTop[q] = ok ? Ph[feat_off(M - 1) + I] : 0.0;
            lamTop[q] = ok ? dP[feat_off(M - 1) + I] : 0.0;
#pragma unroll
            for (int f = 0; f < D; ++f) Gtop[q][f] = ok ? dP[feat_off(M) + int64_t(I) * D + f] : 0.0;
        }
        __syncthreads();
        int cur = 0;
        for (int a = R - 1; a >= 0; --a) {
            const double* const dxa = dx + a * D;
            const double* const phiC = phiB + cur * NLOW;
            double* const phiN = phiB + (cur ^ 1) * NLOW;
            const double* const lamC = lamB + cur * lamlow;
            double* const lamN = lamB + (cur ^ 1) * lamlow;
            double d_[D], gp[D];
#pragma unroll
            for (int f = 0; f < D; ++f) { d_[f] = dxa[f]; gp[f] = 0.0; }
            // ---- phase A: every phi of the state BEFORE the step from the values after it, Phi^{a-1} = Phi^a (x) E(dx)^{-1}: along the chain
            // of an entry's prefixes,  s_0 = cinv[n],  s_j = cinv[n-j] phi_j^a[prefix j] + dx[i_j] s_{j-1}  (old values only: one phase)
            double hnew[LPT], lacc[LPT];
#pragma unroll
            for (int s = 0; s < LPT; ++s) {
                const int n = ln[s];
                hnew[s] = 1.0; lacc[s] = 0.0;
                if (n < 0) continue;
                double h = A.cinv[n];
#pragma unroll
                for (int j = 1; j <= NCH; ++j)
                    if (j <= n) h = fma(A.cinv[n - j], phiC[choff[s][j - 1]], dxa[chcmp[s][j - 1]] * h);
                hnew[s] = h;
                phiN[lphi[s]] = h;                                          // (level 0 stays 1: cinv[0] == 1)
            }
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                double h = A.cinv[M - 1];
#pragma unroll
                for (int j = 1; j <= M - 2; ++j) h = fma(A.cinv[M - 1 - j], phiC[tpre[q][j - 1]], dxa[tcmp[q][j - 1]] * h);
                phiTop[q] = fma(A.cinv[0], phiTop[q], dxa[tcmp[q][M - 2]] * h);
            }
            __syncthreads();
            // The step is P Horner sub-steps  T^(j) = Phi + shift(T^(j+1)) / (j+1),  T^(P) = Phi^{a-1},  T^(0) = Phi^a  (shift(T)_m = T_{m-1} (x) dx).
            // Their adjoints: tau^(0) = lam^a,  tau^(j+1)_n[I] = sum_f tau^(j)_{n+1}[I, f] dx[f] / (j+1),  lam^{a-1} = sum_j tau^(j),
            //                 g += T^(j+1)_n[I] tau^(j)_{n+1}[I, :] / (j+1)  -- one phase per j, a level reading what the level above wrote.
            // T^(jj)_n[I] from the NEW values along the entry's prefixes: sum_k w[jj][k] phi_{n-k}[prefix] dx[i_{n-k+1}] .. dx[i_n], k <= min(P - jj, n)
            auto t_low = [&](int s, int n, int jj) {
                const int kmax = (P - jj) < n ? (P - jj) : n;
                double t = 0.0;
#pragma unroll
                for (int lv = 0; lv <= NCH; ++lv) {
                    const int k = n - lv;
                    if (k < 0 || k > kmax) continue;
                    double val = 1.0, ext = 0.0;                     // the entry's prefix of length lv; the component that extends prefix lv-1 to it
                    if constexpr (true) {
                        if (lv >= 1) {
                            val = lv == n ? hnew[s] : phiN[choff[s][lv >= 1 ? lv - 1 : 0]];
                            ext = dxa[chcmp[s][lv >= 1 ? lv - 1 : 0]];
                        }
                    }
                    t = k == kmax ? A.w[jj][k] * val : fma(A.w[jj][k], val, ext * t);
                }
                return t;
            };
            // ---- phase 0: tau^(0) = lam^a (the top level's is the upstream itself, in registers)
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const int kmax = (P - 1) < (M - 1) ? (P - 1) : (M - 1);
                double t = 0.0;
#pragma unroll
                for (int lv = 0; lv <= M - 1; ++lv) {
                    const int k = M - 1 - lv;
                    if (k > kmax) continue;
                    double val = 1.0, ext = 0.0;
                    if (lv >= 1) {
                        val = lv == M - 1 ? phiTop[q] : phiN[tpre[q][(lv >= 1 && lv <= NCT) ? lv - 1 : 0]];
                        ext = dxa[tcmp[q][lv >= 1 ? lv - 1 : 0]];
                    }
                    t = k == kmax ? A.w[1][k] * val : fma(A.w[1][k], val, ext * t);
                }
                double acc = 0.0;
#pragma unroll
                for (int f = 0; f < D; ++f) {
                    gp[f] = fma(t, Gtop[q][f], gp[f]);
                    acc = fma(Gtop[q][f], d_[f], acc);
                }
                const int I = q * T + tid;
                if (P >= 2 && I < NTOPROWS) tauA[lam_off(M - 1) + I] = acc;       // tau^(1) of level M-1
                lamTop[q] += acc;                                                 // lam^{a-1}: tau^(0) + tau^(1) (tau^(2) of this level is zero)
            }
#pragma unroll
            for (int s = 0; s < LPT; ++s) {
                const int n = ln[s];
                if (n < 0) continue;
                const double* row = n + 1 == M - 1 ? lamM1 + (lrow[s] - lam_off(M - 1)) : lamC + lrow[s];
                const double t = t_low(s, n, 1);
                double acc = 0.0;
#pragma unroll
                for (int f = 0; f < D; ++f) {
                    const double l = row[f];
                    gp[f] = fma(t, l, gp[f]);
                    acc = fma(l, d_[f], acc);
                }
                if (n >= 1) {
                    lacc[s] = lamC[llam[s]] + acc;
                    if (P >= 2) tauA[llam[s]] = acc;
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const int I = q * T + tid;
                if (I < NTOPROWS) lamM1[I] = lamTop[q];                            // (read again in the next step's phase 0)
            }
            // ---- phases 1 .. P-1: level n takes part while the level above still has a tau (n + 1 <= M - j)
            for (int j = 1; j < P && j <= M - 1; ++j) {
                const double* const tin = (j & 1) ? tauA : tauB;
                double* const tout = (j & 1) ? tauB : tauA;
                const double inv = 1.0 / double(j + 1);
#pragma unroll
                for (int s = 0; s < LPT; ++s) {
                    const int n = ln[s];
                    if (n < 0 || n > M - 1 - j) continue;
                    const double* row = tin + lrow[s];
                    const double t = inv * t_low(s, n, j + 1);
                    double acc = 0.0;
#pragma unroll
                    for (int f = 0; f < D; ++f) {
                        const double l = row[f];
                        gp[f] = fma(t, l, gp[f]);
                        acc = fma(l, d_[f], acc);
                    }
                    acc *= inv;
                    if (n >= 1) {
                        lacc[s] += acc;
                        if (j + 1 < P) tout[llam[s]] = acc;                    // tau^(j+1) of this level: read by the level below in the next phase
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int s = 0; s < LPT; ++s)
                if (ln[s] >= 1) lamN[llam[s]] = lacc[s];
            // ---- g_a: per-thread partial sums -> per-wave sums (wave-private tile, FB components per pass) -> slot of this step
            double* const wt = tile + size_t(wave) * FB * 64;
            double* const slot = slots + (size_t(a % SIG_GRAD_TCH) * W + wave) * D;
#pragma unroll
            for (int f0 = 0; f0 < D; f0 += FB) {
#pragma unroll
                for (int f = 0; f < FB; ++f)
                    if (f0 + f < D) wt[f * 64 + lane] = gp[f0 + f];
                // (same wavefront: the LDS unit executes a wave's accesses in order)
                const int fl = lane / GL, sub = lane % GL;
                double s = 0.0;
                if (fl < FB && f0 + fl < D) {
#pragma unroll
                    for (int k = 0; k < 64 / GL; ++k) s += wt[fl * 64 + sub + GL * k];
                }
#pragma unroll
                for (int o = GL / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                if (fl < FB && f0 + fl < D && sub == 0) slot[f0 + fl] = s;
            }
            __syncthreads();                                                // lamN / phiN complete; slots of this step written
            if (a % SIG_GRAD_TCH == 0 || a == 0) {                          // add up the waves' sums of the steps a .. (held ones)
                const int a_hi = (a / SIG_GRAD_TCH) * SIG_GRAD_TCH + SIG_GRAD_TCH - 1 < R - 1 ? (a / SIG_GRAD_TCH) * SIG_GRAD_TCH + SIG_GRAD_TCH - 1 : R - 1;
                for (int e = tid; e < (a_hi - a + 1) * D; e += T) {
                    const int aa = a + e / D, f = e % D;
                    double s = 0.0;
                    for (int w = 0; w < W; ++w) s += slots[(size_t(aa % SIG_GRAD_TCH) * W + w) * D + f];
                    gsum[aa * D + f] = s;
                }
                // (the next step's slot writes come after its first barrier: no race with these reads)
            }
            cur ^= 1;
        }
        __syncthreads();
        // d/dX from d/d(increments): signature_algs.py:25-26 takes differences of the kernel matrix = increments of the sequence here
        double* gx = A.gX + sq * int64_t(A.L) * D;
        if (!A.unit_points) {
            for (int e = tid; e < A.L * D; e += T) {
                const int t = e / D, f = e - t * D;
                double v;
                if (A.difference) v = (t >= 1 ? gsum[(t - 1) * D + f] : 0.0) - (t < R ? gsum[t * D + f] : 0.0);
                else v = gsum[e];
                gx[e] = v;
            }
        } else {                                    // through u = x / |x|:  gx = (gu - u <u, gu>) / |x|
            for (int t = tid; t < A.L; t += T) {
                double gu[D], u[D], ss = 0.0, dot = 0.0;
#pragma unroll
                for (int f = 0; f < D; ++f) {
                    gu[f] = A.difference ? (t >= 1 ? gsum[(t - 1) * D + f] : 0.0) - (t < R ? gsum[t * D + f] : 0.0) : gsum[t * D + f];
                    u[f] = Xn[t * D + f];
                    ss = fma(u[f], u[f], ss);
                }
                const double inv = 1.0 / sqrt(ss);
#pragma unroll
                for (int f = 0; f < D; ++f) { u[f] *= inv; dot = fma(u[f], gu[f], dot); }
#pragma unroll
                for (int f = 0; f < D; ++f) gx[t * D + f] = (gu[f] - u[f] * dot) * inv;
            }
        }
    }
}


}  // namespace gpsig
