// wide_kernels.hpp -- gfx950 kernels of the WIDE-STATE-SPACE route (round 6).
//
// Reference shapes: benchmarks/run_gpsig_benchmarks.py:32 runs every data set with num_lags=1 on time-augmented data, a state space of
// 2 (n_features + 1) columns (gpsig/kernels.py:350): 10 .. 1,928 for 9 of the 16 data sets of benchmarks/datasets.json.  The exact-shape kernels
// keep a lane's operands in registers (Kzx: <= 8 columns, the sequence lattices: <= 32); beyond that the base kernel's matrix of
// kernels.py:226 / :329 / :333 is what the reference says it is, a d-deep contraction -- and north_star puts exactly that on the BLAS
// ("MFMA only for the dense matmuls where it is a true contraction").  Here:
//
//   * rows are AUGMENTED so that one dgemm yields the kernel ARGUMENT of the distance kernels, not just the inner product:
//         left form  [v, -|v|^2/2, 1],  right form [v, 1, -|v|^2/2]:   <left(z), right(x)> = <z, x> - |z|^2/2 - |x|^2/2 = -|z - x|^2 / 2 =: a
//     (kernels.py:765-776 forms the same three terms); RBF (:862-864) is exp(a), the Matern families (:955-993) functions of r = sqrt(max(-2a, 1e-40));
//     the reverse pass needs no norm bookkeeping either: the adjoint of a times the augmented rows IS the chain rule through the norms;
//   * ONE kernel then maps arguments to kappa, takes the increments' difference (kernels.py:329-330), the time difference
//     (signature_algs.py:114) and sweeps the chains of signature_algs.py:118-125 -- the argument array (rows = (sequence, time), columns =
//     (component, endpoint, tensor), tensors fastest: a wavefront reads 512 contiguous bytes per component and step) is read once, nothing else
//     of the (lt, T, N, L) tensor ever exists;
//   * the reverse kernel undoes the chains from their totals (as tvs_grad_tile_kernel.hpp does) and writes the adjoint of the argument array, which
//     two more dgemms contract with the augmented rows of the other side.
// Lane = inducing tensor; levels one after the other (level i owns components i(i-1)/2 .. i(i-1)/2 + i - 1, so every column is read by one level).
#pragma once

#include <hip/hip_runtime.h>

#include "seq_core.hpp"
#include "grad_wave_core.hpp"
#include "fast_exp.hpp"

namespace gpsig {

constexpr int WIDE_MAX_LEVELS = 8;
constexpr int WIDE_MAX_ORDER = 4;       // higher-order chains: repeat counts kept per component inside a time step

struct WideTvsArgs {
    const double* arg;      // (Nc * L, CW): row (n - n0) * L + tau, column (k * E + e) * Tpad + t
    int64_t CW, Tpad, Tn;
    int64_t n0, Nc, N;      // this launch: sequences n0 .. n0 + Nc - 1 of N
    int32_t L, M, kind, difference, sum_levels;
    const double* fx;       // (N, M+1) per-sequence factors or NULL
    const double* w;        // (M+1) level weights or NULL
    double* out;            // forward: (T, N) level sum or (M+1, T, N)
    double* aux;            // (N, lt, Tpad) chain totals: written by the forward kernel, read by the reverse kernel; NULL: neither
    const double* G;        // reverse: upstream gradient, (T, N) of the weighted sum (weighted = 1) or (M+1, T, N) of the levels
    double* W;              // reverse: adjoint of arg, same shape
    double* gfac_part;      // reverse, weighted: (TB, N, M+1) partial sums of dL/dfac over the tensors of a block, or NULL
    int32_t weighted;
    int32_t order;          // > 1: the higher-order chains of signature_algs.py:129-160 (at most WIDE_MAX_ORDER)
};

// The base kernel as a function of the argument a = -|z - x|^2 / 2, described by wavefront-uniform numbers so that the three Matern families share
// one instruction stream:  kappa = P(r') exp(-r'),  r' = c r,  P = 1 + a1 r' + a2 r'^2  (kernels.py:955-993: (c, a1, a2) = (1, 0, 0), (sqrt 3, 1, 0),
// (sqrt 5, 1, 1/3)); RBF (:862-864) is exp(a) (compile-time: no square root, no polynomial).  exp through the 64-entry table of fast_exp.hpp in LDS
// (13 instructions, <= 1.3 ulp) instead of the library routine (~50 with its special cases -- the first form of these kernels spent 10x the
// instructions of the recursion on them).
// Matern-1/2 is not differentiable at coinciding points, and the squared distance of two EQUAL rows out of a dgemm is rounding noise (~1e-16 |x|^2)
// instead of the exact zero the exact-shape kernels get from coordinate differences: such distances count as zero (the clamp of kernels.py:781 then
// gives kappa = 1 and passes no gradient, as it does there; the reference's own float64 value at such a pair is 1e-8 from one -- DESIGN section 5).
constexpr double WIDE_M12_COINCIDE = 1e-13;

struct WideKap {
    const double* tab;      // 2^(j/64) in LDS
    double c2, a1, a2, thr;
};

__device__ __forceinline__ WideKap wide_kap(int kind, const double* tab) {
    WideKap K;
    K.tab = tab;
    K.c2 = kind == BASE_MATERN32 ? 3.0 : (kind == BASE_MATERN52 ? 5.0 : 1.0);
    K.a1 = kind == BASE_MATERN12 ? 0.0 : 1.0;
    K.a2 = kind == BASE_MATERN52 ? 1.0 / 3.0 : 0.0;
    K.thr = kind == BASE_MATERN12 ? WIDE_M12_COINCIDE : 1e-40;
    return K;
}

template <bool RBF>
__device__ __forceinline__ double wide_kappa(const WideKap& K, double a) {
    if constexpr (RBF) {
        return kexp_tab(a, K.tab);
    } else {
        double dist = -2.0 * a;
        dist = dist > K.thr ? dist : 0.0;
        const double rp = sqrt(fmax(dist, 1e-40) * K.c2);              // c * sqrt(max(r^2, 1e-40)): kernels.py:779-781
        return fma(fma(K.a2, rp, K.a1), rp, 1.0) * kexp_tab(-rp, K.tab);
    }
}

// kappa and d kappa / d a  (a = -dist / 2: d/da = -2 d/ddist; r' = c sqrt(-2a): dr'/da = -c^2 / r'; the clamp of kernels.py:781 passes no gradient, as
// grad_core.hpp: base_eval_grad)
template <bool RBF>
__device__ __forceinline__ void wide_kappa_grad(const WideKap& K, double a, double& k, double& dk) {
    if constexpr (RBF) {
        k = kexp_tab(a, K.tab);
        dk = k;
    } else {
        double dist = -2.0 * a;
        dist = dist > K.thr ? dist : 0.0;
        const bool clamped = !(dist > 1e-40);
        const double rp = sqrt(fmax(dist, 1e-40) * K.c2);
        const double e = kexp_tab(-rp, K.tab);
        const double P = fma(fma(K.a2, rp, K.a1), rp, 1.0), dP = fma(2.0 * K.a2, rp, K.a1);
        k = P * e;
        dk = clamped ? 0.0 : (K.c2 / rp) * (P - dP) * e;
    }
}

// the raw arguments of I components at one argument row (r points at the lane's column 0 of that row): all loads first, so that they are in flight together
template <int I, int E>
__device__ __forceinline__ void wide_load_row(const double* __restrict__ r, int k0, int64_t Tpad, double (&raw)[I * E]) {
#pragma unroll
    for (int q = 0; q < I * E; ++q) raw[q] = r[(int64_t(k0) * E + q) * Tpad];
}
// the value of component j from them: kappa, or the difference of its two points' (kernels.py:330)
template <int I, int E, bool RBF>
__device__ __forceinline__ double wide_val(const double (&raw)[I * E], int j, const WideKap& K) {
    if constexpr (E == 2) return wide_kappa<RBF>(K, raw[2 * j + 1]) - wide_kappa<RBF>(K, raw[2 * j]);
    else return wide_kappa<RBF>(K, raw[j]);
}

// the chains of ONE level (I components from k0) of one (tensor, sequence) pair: signature_algs.py:118-125 as one sweep; the arguments of the next
// step are requested before the current step is evaluated
// One time step of the HIGHER-ORDER chains of a level (signature_algs.py:147-158): with U_j the running totals (what first order keeps),
//   r_0 = [m_0],   r_j[0] = m_j U_{j-1},   r_j[l] = m_j r_{j-1}[l-1] / (l+1)  (l < min(j+1, order): the SAME time step),   U_j += sum_l r_j[l].
template <int I>
__device__ __forceinline__ void wide_ho_step(const double (&dk)[I], int order, double (&u)[I]) {
    constexpr int O = WIDE_MAX_ORDER;
    double rp[O], uold = u[0];
    rp[0] = dk[0];
    u[0] += dk[0];
#pragma unroll
    for (int j = 1; j < I; ++j) {
        double rc[O], tot = dk[j] * uold;
        rc[0] = tot;
#pragma unroll
        for (int l = 1; l < O; ++l) {
            rc[l] = (l <= j && l < order) ? (dk[j] * (1.0 / double(l + 1))) * rp[l - 1] : 0.0;
            tot += rc[l];
        }
        uold = u[j];
        u[j] += tot;
#pragma unroll
        for (int l = 0; l < O; ++l) rp[l] = rc[l];
    }
}

template <int I, int E, bool RBF>
__device__ __forceinline__ void wide_chain_fwd(const double* __restrict__ col, int64_t CW, int64_t Tpad, int k0, int L, int difference, int order, const WideKap& K,
                                               double (&u)[I]) {
#pragma unroll
    for (int j = 0; j < I; ++j) u[j] = 0.0;
    double prev[I], cur[I * E];
    const double* r = col;
    int steps = L;
    wide_load_row<I, E>(r, k0, Tpad, cur);
    if (difference) {                                                              // signature_algs.py:114
#pragma unroll
        for (int j = 0; j < I; ++j) prev[j] = wide_val<I, E, RBF>(cur, j, K);
        r += CW;
        steps = L - 1;
        if (steps > 0) wide_load_row<I, E>(r, k0, Tpad, cur);
    }
    for (int s = 0; s < steps; ++s) {
        double nxt[I * E];
        r += CW;
        if (s + 1 < steps) wide_load_row<I, E>(r, k0, Tpad, nxt);
        double dk[I];
#pragma unroll
        for (int j = 0; j < I; ++j) {
            const double v = wide_val<I, E, RBF>(cur, j, K);
            dk[j] = difference ? v - prev[j] : v;
            prev[j] = v;
        }
        if (I > 1 && order > 1) {
            wide_ho_step<I>(dk, order, u);
        } else {
#pragma unroll
            for (int j = I - 1; j >= 1; --j) u[j] = fma(dk[j], u[j - 1], u[j]);      // :120-124 (old values below)
            u[0] += dk[0];
        }
#pragma unroll
        for (int q = 0; q < I * E; ++q) cur[q] = nxt[q];
    }
}

template <int E, bool RBF>
__device__ __forceinline__ void wide_level_fwd(int i, const double* col, const WideTvsArgs& A, const WideKap& K, double* u /* [i] */) {
    const int k0 = i * (i - 1) / 2;
#define GPSIG_WIDE_CASE(I_)                                                                          \
    case I_: {                                                                                       \
        double v[I_];                                                                                \
        wide_chain_fwd<I_, E, RBF>(col, A.CW, A.Tpad, k0, A.L, A.difference, A.order, K, v);                  \
        _Pragma("unroll") for (int j = 0; j < I_; ++j) u[j] = v[j];                                 \
    } break;
    switch (i) {
        GPSIG_WIDE_CASE(1) GPSIG_WIDE_CASE(2) GPSIG_WIDE_CASE(3) GPSIG_WIDE_CASE(4)
        GPSIG_WIDE_CASE(5) GPSIG_WIDE_CASE(6) GPSIG_WIDE_CASE(7) GPSIG_WIDE_CASE(8)
        default: break;
    }
#undef GPSIG_WIDE_CASE
}

// grid: (Tpad / 64, sequences of the chunk (grid-stride), levels); block: one wavefront, lane = tensor.  A launch of few sequences is bound by the
// serial sweep of one chain: the levels of a (tensor, sequence) pair go to different workgroups (blockIdx.z + 1 = level), each leaving its chain totals
// in aux (N, lt, Tpad) -- tensors fastest --, and wide_tvs_epilogue_kernel forms the outputs from them.
template <int E, bool RBF>
__global__ void __launch_bounds__(64) wide_tvs_fwd_kernel(const WideTvsArgs A) {
    __shared__ double etab[EXP_TAB_N];
    exp_tab_fill(etab, threadIdx.x, 64);
    __syncthreads();
    const WideKap K = wide_kap(A.kind, etab);
    const int64_t t = int64_t(blockIdx.x) * 64 + threadIdx.x;
    const int M = A.M, lt = M * (M + 1) / 2;
    const int i = blockIdx.z + 1, k0 = i * (i - 1) / 2;
    for (int64_t nl = blockIdx.y; nl < A.Nc; nl += gridDim.y) {
        const int64_t n = A.n0 + nl;
        const double* col = A.arg + nl * int64_t(A.L) * A.CW + t;
        double u[WIDE_MAX_LEVELS];
        wide_level_fwd<E, RBF>(i, col, A, K, u);
        for (int j = 0; j < i; ++j) A.aux[(n * lt + k0 + j) * A.Tpad + t] = u[j];
    }
}

// out[t][n] = fac_0 + sum_i fac_i K_i  or  out[i][t][n] = fac_i K_i  (fac = fx[n][i] * w[i]: kernels.py:572-588), K_i = the total of level i's last chain
// (signature_algs.py:125).  One thread per (t, n) of the chunk, tensors fastest.
__global__ void wide_tvs_epilogue_kernel(const WideTvsArgs A) {
    const int M = A.M, lt = M * (M + 1) / 2;
    const int64_t total = A.Nc * A.Tn;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t t = idx % A.Tn, n = A.n0 + idx / A.Tn;
        double acc = 0.0;
        for (int i = 0; i <= M; ++i) {
            double f = A.fx ? A.fx[n * (M + 1) + i] : 1.0;
            if (A.w) f *= A.w[i];
            const double lev = i == 0 ? 1.0 : A.aux[(n * lt + i * (i - 1) / 2 + i - 1) * A.Tpad + t];       // signature_algs.py:116 / :125
            if (A.sum_levels) acc = fma(lev, f, acc);
            else A.out[(int64_t(i) * A.Tn + t) * A.N + n] = lev * f;
        }
        if (A.sum_levels) A.out[t * A.N + n] = acc;
    }
}

// ---- reverse pass ------------------------------------------------------------------------------------------------------------------
// With u_j the chain values BEFORE a step and W_j = dL/du_j after it:  dL/dm_j = u_{j-1} W_j,  W_{j-1} += m_j W_j,  and the chain is undone
// by u_j <- u_j - m_j u_{j-1} (tvs_grad_tile_kernel.hpp, grad_ho_kernels.hpp: chain_levels_grad_kernel).  m_j[tau] = v_j[tau + 1] - v_j[tau]
// (signature_algs.py:114), so the adjoint of the VALUE row rho is g[rho - 1] - g[rho]; times d kappa / d a of each endpoint it is the adjoint of the
// argument.  u: the level's totals (destroyed).  c: the level's upstream gradient.  Columns of tensors beyond Tn get zeros (valid = false).
template <int I, int E, bool RBF>
__device__ __forceinline__ void wide_chain_bwd(const double* __restrict__ col, double* __restrict__ wcol, int64_t CW, int64_t Tpad, int k0, int L,
                                               int difference, int order, const WideKap& K, double (&u)[I], double c, bool valid) {
    double wv[I];
#pragma unroll
    for (int j = 0; j < I; ++j) wv[j] = 0.0;
    auto eval = [&](const double (&raw)[I * E], double (&v)[I], double (&d)[I][E]) {
#pragma unroll
        for (int j = 0; j < I; ++j) {
            if constexpr (E == 2) {
                double k1, d1, k0v, d0;
                wide_kappa_grad<RBF>(K, raw[2 * j + 1], k1, d1);
                wide_kappa_grad<RBF>(K, raw[2 * j], k0v, d0);
                v[j] = k1 - k0v; d[j][1] = d1; d[j][0] = -d0;
            } else {
                wide_kappa_grad<RBF>(K, raw[j], v[j], d[j][0]);
            }
        }
    };
    auto store = [&](double* __restrict__ wr, const double (&g)[I], const double (&d)[I][E]) {
#pragma unroll
        for (int j = 0; j < I; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) wr[((k0 + j) * E + e) * Tpad] = valid ? g[j] * d[j][e] : 0.0;
    };
    // higher-order chains: the step's repeat-count vectors are rebuilt from the totals BEFORE the step (U_j - sum_l r_j[l], ascending in j), then the
    // adjoints run down: dL/dr_j[l] = W_j + m_{j+1} / (l+2) dL/dr_{j+1}[l+1],  dL/dm_j = dL/dr_j[0] U_{j-1} + sum_{l>=1} dL/dr_j[l] r_{j-1}[l-1] / (l+1),
    // W_{j-1} += m_j dL/dr_j[0]   (W_j = dL/dU_j after the step; first order: one repeat count)
    auto undo_ho = [&](const double (&dk)[I], double (&g)[I]) {
        constexpr int O = WIDE_MAX_ORDER;
        double r[I][O], ub[I];
        ub[0] = u[0] - dk[0];
#pragma unroll
        for (int l = 0; l < O; ++l) r[0][l] = l == 0 ? dk[0] : 0.0;
#pragma unroll
        for (int j = 1; j < I; ++j) {
            double tot = dk[j] * ub[j - 1];
            r[j][0] = tot;
#pragma unroll
            for (int l = 1; l < O; ++l) {
                r[j][l] = (l <= j && l < order) ? (dk[j] * (1.0 / double(l + 1))) * r[j - 1][l - 1] : 0.0;
                tot += r[j][l];
            }
            ub[j] = u[j] - tot;
        }
        double gn[O], add[I];                             // dL/dr_{j+1}[.];  m_j dL/dr_j[0], added to W_{j-1} once the step's adjoints are through
#pragma unroll
        for (int l = 0; l < O; ++l) gn[l] = 0.0;
#pragma unroll
        for (int j = I - 1; j >= 0; --j) {
            const double wj = (j == I - 1) ? c : wv[j + 1 < I ? j + 1 : j];
            double gr[O];
#pragma unroll
            for (int l = 0; l < O; ++l) {
                const bool live = l <= j && l < order;
                const bool up = j + 1 < I && l + 1 < O && l + 1 <= j + 1 && l + 1 < order;
                gr[l] = live ? wj + (up ? (dk[j + 1 < I ? j + 1 : j] * (1.0 / double(l + 2))) * gn[l + 1 < O ? l + 1 : l] : 0.0) : 0.0;
            }
            double gd = gr[0] * (j >= 1 ? ub[j >= 1 ? j - 1 : 0] : 1.0);
#pragma unroll
            for (int l = 1; l < O; ++l)
                if (j >= 1) gd = fma(gr[l] * (1.0 / double(l + 1)), r[j >= 1 ? j - 1 : 0][l - 1], gd);
            g[j] = gd;
            add[j] = dk[j] * gr[0];
#pragma unroll
            for (int l = 0; l < O; ++l) gn[l] = gr[l];
        }
#pragma unroll
        for (int j = 1; j < I; ++j) wv[j] += add[j];
#pragma unroll
        for (int j = 0; j < I; ++j) u[j] = ub[j];
    };
    auto undo1 = [&](const double (&dk)[I], double (&g)[I]) {
        double below = 1.0;
#pragma unroll
        for (int j = 0; j < I; ++j) {
            const double wnext = (j == I - 1) ? c : wv[j + 1 < I ? j + 1 : j];
            g[j] = below * wnext;
            u[j] = fma(-dk[j], below, u[j]);
            below = u[j];
            if (j >= 1) wv[j] = fma(dk[j], wnext, wv[j]);
        }
    };
    auto undo = [&](const double (&dk)[I], double (&g)[I]) {
        if (I > 1 && order > 1) undo_ho(dk, g);
        else undo1(dk, g);
    };
    // (the arguments of the row after next are requested before a row is evaluated: all its loads in flight together)
    const double* r = col + int64_t(L - 1) * CW;
    double* wr = wcol + int64_t(L - 1) * CW;
    double raw[I * E], rnx[I * E];
    wide_load_row<I, E>(r, k0, Tpad, raw);
    if (difference) {
        double nv[I], nd[I][E], gprev[I];
        eval(raw, nv, nd);
#pragma unroll
        for (int j = 0; j < I; ++j) gprev[j] = 0.0;
        if (L >= 2) wide_load_row<I, E>(r - CW, k0, Tpad, raw);
        for (int s = L - 2; s >= 0; --s) {
            r -= CW;
            if (s >= 1) wide_load_row<I, E>(r - CW, k0, Tpad, rnx);
            double cv[I], cd[I][E], dk[I], g[I], gv[I];
            eval(raw, cv, cd);
#pragma unroll
            for (int j = 0; j < I; ++j) dk[j] = nv[j] - cv[j];
            undo(dk, g);
#pragma unroll
            for (int j = 0; j < I; ++j) gv[j] = g[j] - gprev[j];
            store(wr, gv, nd);                                                         // row s + 1 is final
            wr -= CW;
#pragma unroll
            for (int j = 0; j < I; ++j) {
                gprev[j] = g[j]; nv[j] = cv[j];
#pragma unroll
                for (int e = 0; e < E; ++e) nd[j][e] = cd[j][e];
            }
#pragma unroll
            for (int q = 0; q < I * E; ++q) raw[q] = rnx[q];
        }
        double gv[I];
#pragma unroll
        for (int j = 0; j < I; ++j) gv[j] = -gprev[j];
        store(wr, gv, nd);                                                             // row 0
    } else {
        for (int s = L - 1; s >= 0; --s, r -= CW, wr -= CW) {
            if (s >= 1) wide_load_row<I, E>(r - CW, k0, Tpad, rnx);
            double cv[I], cd[I][E], g[I];
            eval(raw, cv, cd);
            undo(cv, g);
            store(wr, g, cd);
#pragma unroll
            for (int q = 0; q < I * E; ++q) raw[q] = rnx[q];
        }
    }
}

__device__ __forceinline__ double wide_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int E, bool RBF>
__global__ void __launch_bounds__(64) wide_tvs_bwd_kernel(const WideTvsArgs A) {
    __shared__ double etab[EXP_TAB_N];
    exp_tab_fill(etab, threadIdx.x, 64);
    __syncthreads();
    const WideKap K = wide_kap(A.kind, etab);
    const int64_t t = int64_t(blockIdx.x) * 64 + threadIdx.x;
    const bool valid = t < A.Tn;
    const int M = A.M, lt = M * (M + 1) / 2;
    const int i = blockIdx.z + 1, k0 = i * (i - 1) / 2;          // one level per workgroup (as the forward kernel)
    for (int64_t nl = blockIdx.y; nl < A.Nc; nl += gridDim.y) {
        const int64_t n = A.n0 + nl;
        const double* col = A.arg + nl * int64_t(A.L) * A.CW + t;
        double* wcol = A.W + nl * int64_t(A.L) * A.CW + t;
        const double gsum = (A.weighted && valid) ? A.G[t * A.N + n] : 0.0;
        if (A.gfac_part && i == 1) {                                                   // level 0 == 1: dL/dfac[n][0] = sum_t G[t][n]
            const double s = wide_wave_sum(gsum);
            if (threadIdx.x == 0) A.gfac_part[(int64_t(blockIdx.x) * A.N + n) * (M + 1)] = s;
        }
        double f = A.fx ? A.fx[n * (M + 1) + i] : 1.0;
        if (A.w) f *= A.w[i];
        const double c = A.weighted ? gsum * f : (valid ? A.G[(int64_t(i) * A.Tn + t) * A.N + n] * f : 0.0);
        double u[WIDE_MAX_LEVELS];
        if (A.aux) {
            for (int j = 0; j < i; ++j) u[j] = A.aux[(n * lt + k0 + j) * A.Tpad + t];
        } else {
            wide_level_fwd<E, RBF>(i, col, A, K, u);
        }
        if (A.gfac_part) {
            double ui = 0.0;
#pragma unroll
            for (int j = 0; j < WIDE_MAX_LEVELS; ++j)
                if (j == i - 1) ui = u[j];
            const double s = wide_wave_sum(gsum * ui);
            if (threadIdx.x == 0) A.gfac_part[(int64_t(blockIdx.x) * A.N + n) * (M + 1) + i] = s;
        }
#define GPSIG_WIDE_CASE(I_)                                                                                  \
    case I_: {                                                                                               \
        double v[I_];                                                                                        \
        _Pragma("unroll") for (int j = 0; j < I_; ++j) v[j] = u[j];                                         \
        wide_chain_bwd<I_, E, RBF>(col, wcol, A.CW, A.Tpad, k0, A.L, A.difference, A.order, K, v, c, valid);          \
    } break;
        switch (i) {
            GPSIG_WIDE_CASE(1) GPSIG_WIDE_CASE(2) GPSIG_WIDE_CASE(3) GPSIG_WIDE_CASE(4)
            GPSIG_WIDE_CASE(5) GPSIG_WIDE_CASE(6) GPSIG_WIDE_CASE(7) GPSIG_WIDE_CASE(8)
            default: break;
        }
#undef GPSIG_WIDE_CASE
    }
}

// gfac[n][m] = sum over the tensor blocks of the partial sums (a fixed order: the result does not depend on the schedule)
__global__ void wide_gfac_reduce_kernel(const double* __restrict__ part, int TB, int64_t NM, double* __restrict__ gfac) {
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < NM; idx += int64_t(gridDim.x) * blockDim.x) {
        double s = 0.0;
        for (int b = 0; b < TB; ++b) s += part[int64_t(b) * NM + idx];
        gfac[idx] = s;
    }
}

// ---- augmented rows ------------------------------------------------------------------------------------------------------------------
// dst row r (DA = d + 2 columns): the d values of the source row, then (-|v|^2/2, 1) [left form] or (1, -|v|^2/2) [right form].
// Tensor rows are reordered on the way: dst row (k * E + e) * Tpad + t  <-  src row (k * Tn + t) * E + e (the caller's (lt, T, E, d) array), scaled by
// lengthscales / lag weights where P.has_ls (kernels.py:367-398); rows of tensors beyond Tn are zero.  lt == 0: rows as they come (sequences, already scaled).
// One wavefront per row.
__global__ void __launch_bounds__(64) wide_aug_rows_kernel(const double* __restrict__ src, int64_t rows, int d, int right, int lt, int64_t Tn, int64_t Tpad,
                                                           int E, ScaleParams P, double* __restrict__ dst) {
    const int DA = d + 2;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const double* s = nullptr;
        if (lt > 0) {
            const int64_t t = r % Tpad;
            const int ke = int(r / Tpad), k = ke / E, e = ke - k * E;
            if (t < Tn) s = src + ((int64_t(k) * Tn + t) * E + e) * d;
        } else {
            s = src + r * d;
        }
        double ss = 0.0;
        for (int f = threadIdx.x; f < d; f += 64) {
            double v = 0.0;
            if (s) {
                v = s[f];
                if (lt > 0 && P.has_ls) {
                    const int lag = f / P.d_in, f0 = f - lag * P.d_in;
                    v = v / P.lsv(f0);
                    if (P.num_lags > 0) v = v * P.gamma[lag];
                }
            }
            dst[r * DA + f] = v;
            ss = fma(v, v, ss);
        }
        ss = wide_wave_sum(ss);
        if (threadIdx.x == 0) {
            const double h = s ? -0.5 * ss : 0.0, one = s ? 1.0 : 0.0;
            dst[r * DA + d] = right ? one : h;
            dst[r * DA + d + 1] = right ? h : one;
        }
    }
}

// The chain rule through the augmentation: g[f] = ga[f] - ga[norm column] * v[f]   (d(-|v|^2/2)/dv = -v; the constant column carries nothing).
// Tensor rows go back to the caller's (lt, T, E, d) order; sequences as they come.  One thread per output element.
__global__ void wide_unaug_rows_kernel(const double* __restrict__ ga, const double* __restrict__ va, int64_t rows_out, int d, int right, int lt, int64_t Tn,
                                       int64_t Tpad, int E, double* __restrict__ g) {
    const int DA = d + 2, nc = right ? d + 1 : d;
    const int64_t total = rows_out * d;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int f = int(idx % d);
        const int64_t ro = idx / d;
        int64_t r = ro;
        if (lt > 0) {
            const int e = int(ro % E);
            const int64_t t = (ro / E) % Tn;
            const int k = int(ro / (int64_t(E) * Tn));
            r = (int64_t(k) * E + e) * Tpad + t;
        }
        g[idx] = fma(-ga[r * DA + nc], va[r * DA + f], ga[r * DA + f]);
    }
}


// =====================================================================================================================================
// Sequence lattices from kernel-argument lattices in memory (the level diagonals of kernels.py:188-205, the sequence Grams of :208-237
// beyond the exact-shape kernels' columns): the sweeps of grad_wave_core.hpp -- one wavefront per lattice, lane lam owns C consecutive columns
// and works on row t - lam at step t, row prefixes handed to the right neighbour by one DPP shift -- fed by loads of the argument rows instead
// of point rows.  arg lattice of pair p = (i, j): base (p / N2) * si + (p % N2) * sj, row stride ld (a batched product: N2 = 1, si = L1 * L2, ld = L2;
// one product of all points: si = L1 * ld, sj = L2, ld = N2 * L2).
struct WideLatArgs {
    const double* arg;
    int64_t ld, si, sj, N2;
    int64_t P, p0, Ptot;        // this launch: lattices p0 .. p0 + P - 1 of Ptot (the level arrays' pair axis)
    int32_t L1, L2, M, kind, difference;
    double* out;                // forward: (M+1, Ptot) level values
    const double* G;            // reverse: (M+1, Ptot) upstream gradient, lattice pg of this launch at  (pg / N2) * g_i + (pg % N2) * g_j
    int64_t g_i, g_j;
    double* scratch;            // reverse: per group (M-1) * TF * 64 * C doubles (forward Q's, as grad_wave_kernel.hpp)
    double* lam;                // reverse: Lam = dL/ddM, (P, R1, R2) row-major
    int32_t ngroups;
};

// argument rows (and, in the backward sweep, stored prefix rows) in flight ahead of a step: a step of few columns is shorter than a memory access
constexpr int wide_lat_pf(int C) { return C >= 8 ? 1 : (C == 4 ? 2 : 4); }

template <int C, bool RBF>
struct WideLatDm {
    double rd[C];
    int nvalid, b0, L2, diff;
    WideKap K;
    int64_t ld;
    const double* lat;

    __device__ __forceinline__ void load(int r, bool ok, double (&raw)[C + 1]) const {
        const double* p = lat + int64_t(r) * ld + b0;
#pragma unroll
        for (int c = 0; c <= C; ++c) raw[c] = (ok && b0 + c < L2 && (c < C || diff)) ? p[c] : 0.0;
    }
    // kappa differences along the row (diff) or kappa itself (no differences: signature_algs.py:18-19 with difference=False)
    __device__ __forceinline__ void map(const double (&raw)[C + 1], double (&out)[C]) const {
        double k[C + 1];
#pragma unroll
        for (int c = 0; c <= C; ++c) k[c] = (c < C || diff) ? wide_kappa<RBF>(K, raw[c]) : 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = diff ? k[c + 1] - k[c] : k[c];
    }
    __device__ __forceinline__ void prime(const double (&raw)[C + 1]) { map(raw, rd); }
    // one lattice row from the raw arguments of its new point row (forward: a + 1, backward: a; no differences: a)
    __device__ __forceinline__ void row(const double (&raw)[C + 1], bool forward, double (&dm)[C]) {
        double nd[C];
        map(raw, nd);
#pragma unroll
        for (int c = 0; c < C; ++c) {               // (selects, no branches: the arrays stay in registers)
            const double o = rd[c];
            const double df = forward ? nd[c] - o : o - nd[c];
            const double v = diff ? df : nd[c];
            rd[c] = nd[c];
            dm[c] = c < nvalid ? v : 0.0;
        }
    }
};

__device__ __forceinline__ double wide_from_left(double v) {       // lane l <- lane l-1, 0 into lane 0
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);     // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wide_from_right(double v) {      // lane l <- lane l+1, 0 into lane 63
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);     // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Level values of P lattices: one wavefront per lattice (grid-stride).  K_m (m < M) = Q_m at the last cell, K_M = the sum of the row totals of R_M:
// both arrive at lane 63 (columns beyond the lattice pass the row prefixes on unchanged).
// NW > 1: NW wavefronts per lattice (a workgroup), lane lam = threadIdx.x of 64 NW; the hand-over across a wavefront boundary goes through LDS (two buffers
// alternating by the step's parity, one barrier per step).  For a FEW long lattices (the level diagonals of a minibatch: 50 lattices of 499 x 499 keep
// 50 of 1,024 SIMDs busy with eight columns per lane) -- more steps (R1 + 64 NW - 1), an eighth of the work per step, NW times the wavefronts.
template <int C, int LQ, bool RBF, int NW = 1>
__global__ void __launch_bounds__(64 * NW) wide_lattice_fwd_kernel(const WideLatArgs A) {
    __shared__ double etab[EXP_TAB_N];
    __shared__ double xw[2][NW][LQ + 2];
    exp_tab_fill(etab, threadIdx.x, 64 * NW);
    __syncthreads();
    constexpr int GL = 64 * NW, PF = wide_lat_pf(C);
    const int lam = threadIdx.x, M = A.M, wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int dr = A.difference ? 1 : 0, R1 = A.L1 - dr, R2 = A.L2 - dr, TF = R1 + GL - 1;
    for (int64_t pp = blockIdx.x; pp < A.P; pp += gridDim.x) {
        if constexpr (NW > 1) {
            __syncthreads();
            if (ln == 63) {
#pragma unroll
                for (int m = 0; m < LQ + 2; ++m) xw[0][wv][m] = xw[1][wv][m] = 0.0;
            }
            __syncthreads();
        }
        const int64_t pg = A.p0 + pp;
        WideLatDm<C, RBF> dmg;
        dmg.lat = A.arg + (pg / A.N2) * A.si + (pg % A.N2) * A.sj;
        dmg.ld = A.ld; dmg.b0 = C * lam; dmg.L2 = A.L2; dmg.K = wide_kap(A.kind, etab); dmg.diff = dr;
        { const int nv = R2 - C * lam; dmg.nvalid = nv < 0 ? 0 : (nv > C ? C : nv); }
        WaveFwd<C, LQ> fw;
        fw.reset();
        // the argument rows of the coming PF steps are in flight while a step computes (a launch of a few lattices is one wavefront per
        // SIMD: nothing else hides the latency of these strided loads)
        double raw[PF][C + 1];
        if (dr) { dmg.load(0, true, raw[0]); dmg.prime(raw[0]); }
#pragma unroll
        for (int u = 0; u < PF; ++u) { const int r = u - lam + dr; dmg.load(r, r >= 0 && r < A.L1, raw[u]); }
        double kM = 0.0;
        for (int t0 = 0; t0 < TF; t0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 + u;
                if (t >= TF) break;
                double cin[LQ + 2], cur[C + 1];
                cin[0] = 0.0;
#pragma unroll
                for (int m = 1; m < LQ + 2; ++m) {
                    cin[m] = wide_from_left(fw.sout[m]);
                    if constexpr (NW > 1) {
                        const double x = xw[t & 1][wv > 0 ? wv - 1 : 0][m];
                        cin[m] = (ln == 0 && wv > 0) ? x : cin[m];
                    }
                }
#pragma unroll
                for (int c = 0; c <= C; ++c) cur[c] = raw[u][c];
                const int a = t - lam;
                { const int r = a + PF + dr; dmg.load(r, r >= 0 && r < A.L1, raw[u]); }
                if (a >= 0 && a < R1) {
                    double dm[C];
                    dmg.row(cur, true, dm);
                    fw.step(dm, cin, M);
#pragma unroll
                    for (int m = 1; m <= LQ + 1; ++m)
                        if (m == M) kM += fw.sout[m];
                }
                if constexpr (NW > 1) {
                    if (ln == 63) {
#pragma unroll
                        for (int m = 1; m < LQ + 2; ++m) xw[(t + 1) & 1][wv][m] = fw.sout[m];
                    }
                    __syncthreads();
                }
            }
        }
        if (lam == GL - 1) {
            A.out[pg] = 1.0;                                                       // signature_algs.py:20
#pragma unroll
            for (int m = 1; m <= LQ; ++m)
                if (m < M) A.out[int64_t(m) * A.Ptot + pg] = fw.q[m - 1][C - 1];
            A.out[int64_t(M) * A.Ptot + pg] = kM;
        }
    }
}

// Both sweeps (grad_wave_kernel.hpp: seq_grad_wave_kernel with the argument lattice in place of the point rows): Lam[a][b] = dL/ddM[a][b] out.
// grid: ngroups workgroups of one wavefront; a group's pairs one after the other through its scratch slot.
template <int C, int LQ, bool RBF, int NW = 1>
__global__ void __launch_bounds__(64 * NW) wide_lattice_bwd_kernel(const WideLatArgs A) {
    __shared__ double etab[EXP_TAB_N];
    __shared__ double xw[2][NW][LQ + 2];
    exp_tab_fill(etab, threadIdx.x, 64 * NW);
    __syncthreads();
    constexpr int GL = 64 * NW, PF = wide_lat_pf(C);
    const int lam = threadIdx.x, M = A.M, wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int dr = A.difference ? 1 : 0, R1 = A.L1 - dr, R2 = A.L2 - dr, TF = R1 + GL - 1;
    double* scr = A.scratch + size_t(blockIdx.x) * size_t(M - 1) * TF * GL * C;
    auto slot = [&](int m, int tf, int l, int c) -> double& { return scr[((size_t(m) * TF + tf) * GL + l) * C + c]; };
    auto xw_clear = [&]() {
        if constexpr (NW > 1) {
            __syncthreads();
            if (ln == 63) {
#pragma unroll
                for (int m = 0; m < LQ + 2; ++m) xw[0][wv][m] = xw[1][wv][m] = 0.0;
            }
            __syncthreads();
        }
    };
    for (int64_t pp = blockIdx.x; pp < A.P; pp += gridDim.x) {
        xw_clear();
        const int64_t pg = A.p0 + pp;
        WideLatDm<C, RBF> dmg;
        dmg.lat = A.arg + (pg / A.N2) * A.si + (pg % A.N2) * A.sj;
        dmg.ld = A.ld; dmg.b0 = C * lam; dmg.L2 = A.L2; dmg.K = wide_kap(A.kind, etab); dmg.diff = dr;
        { const int nv = R2 - C * lam; dmg.nvalid = nv < 0 ? 0 : (nv > C ? C : nv); }
        double clev[LQ + 2];
#pragma unroll
        for (int p = 0; p < LQ + 2; ++p) clev[p] = (p >= 1 && p <= M) ? A.G[int64_t(p) * A.Ptot + (pg / A.N2) * A.g_i + (pg % A.N2) * A.g_j] : 0.0;
        // ---- forward sweep, Q's of levels < M to the scratch slot
        {
            WaveFwd<C, LQ> fw;
            fw.reset();
            double raw[PF][C + 1];
            if (dr) { dmg.load(0, true, raw[0]); dmg.prime(raw[0]); }
#pragma unroll
            for (int u = 0; u < PF; ++u) { const int r = u - lam + dr; dmg.load(r, r >= 0 && r < A.L1, raw[u]); }
            for (int t0 = 0; t0 < TF; t0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int t = t0 + u;
                    if (t >= TF) break;
                    double cin[LQ + 2], cur[C + 1];
                    cin[0] = 0.0;
#pragma unroll
                    for (int m = 1; m < LQ + 2; ++m) {
                        cin[m] = wide_from_left(fw.sout[m]);
                        if constexpr (NW > 1) {
                            const double x = xw[t & 1][wv > 0 ? wv - 1 : 0][m];
                            cin[m] = (ln == 0 && wv > 0) ? x : cin[m];
                        }
                    }
#pragma unroll
                    for (int c = 0; c <= C; ++c) cur[c] = raw[u][c];
                    const int a = t - lam;
                    { const int r = a + PF + dr; dmg.load(r, r >= 0 && r < A.L1, raw[u]); }
                    if (a >= 0 && a < R1) {
                        double dm[C];
                        dmg.row(cur, true, dm);
                        fw.step(dm, cin, M);
#pragma unroll
                        for (int m = 0; m < LQ; ++m)
                            if (m < M - 1) {
#pragma unroll
                                for (int c = 0; c < C; ++c) slot(m, t, lam, c) = fw.q[m][c];
                            }
                    }
                    if constexpr (NW > 1) {
                        if (ln == 63) {
#pragma unroll
                            for (int m = 1; m < LQ + 2; ++m) xw[(t + 1) & 1][wv][m] = fw.sout[m];
                        }
                        __syncthreads();
                    }
                }
            }
        }
        __threadfence();        // the backward sweep reads what other lanes of this wavefront (workgroup) stored
        xw_clear();
        // ---- backward sweep
        {
            WaveBwd<C, LQ> bw;
            bw.reset();
            double rawr[PF][C + 1], qr[PF][LQ][C];
            if (dr) { dmg.load(R1, true, rawr[0]); dmg.prime(rawr[0]); }
            double* lamrow = A.lam + size_t(pp) * R1 * R2;
            auto fetch_q = [&](int a, double (&q)[LQ][C]) {
                const int tf = a - 1 + lam;
#pragma unroll
                for (int m = 0; m < LQ; ++m)
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        double v = 0.0;
                        if (m < M - 1 && a > 0 && a < R1) {
                            if (c > 0) v = slot(m, tf, lam, c - 1);
                            else if (lam > 0) v = slot(m, tf - 1, lam - 1, C - 1);
                        }
                        q[m][c] = v;
                    }
            };
            // the rows of the coming PF steps (arguments and stored prefixes) are in flight while a step computes
#pragma unroll
            for (int v = 0; v < PF; ++v) {
                const int r = R1 - 1 - (v - (GL - 1 - lam));
                dmg.load(r, r >= 0 && r < R1, rawr[v]);
                fetch_q(r, qr[v]);
            }
            for (int u0 = 0; u0 < TF; u0 += PF) {
#pragma unroll
                for (int v = 0; v < PF; ++v) {
                    const int u = u0 + v;
                    if (u >= TF) break;
                    double sin[LQ], raw[C + 1], qcur[LQ][C];
#pragma unroll
                    for (int p = 0; p < LQ; ++p) {
                        sin[p] = wide_from_right(bw.svout[p]);
                        if constexpr (NW > 1) {
                            const double x = xw[u & 1][wv + 1 < NW ? wv + 1 : wv][p];
                            sin[p] = (ln == 63 && wv + 1 < NW) ? x : sin[p];
                        }
                    }
                    const int a = R1 - 1 - (u - (GL - 1 - lam));
#pragma unroll
                    for (int c = 0; c <= C; ++c) raw[c] = rawr[v][c];
#pragma unroll
                    for (int m = 0; m < LQ; ++m)
#pragma unroll
                        for (int c = 0; c < C; ++c) qcur[m][c] = qr[v][m][c];
                    { const int r = a - PF; dmg.load(r, r >= 0 && r < R1, rawr[v]); fetch_q(r, qr[v]); }
                    if (a >= 0 && a < R1) {
                        double dm[C], lv[C];
                        dmg.row(raw, false, dm);
                        bw.step(dm, clev, qcur, sin, M, lv);
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            if (c < dmg.nvalid) lamrow[size_t(a) * R2 + C * lam + c] = lv[c];
                    }
                    if constexpr (NW > 1) {
                        if (ln == 0) {
#pragma unroll
                            for (int p = 0; p < LQ; ++p) xw[(u + 1) & 1][wv][p] = bw.svout[p];
                        }
                        __syncthreads();
                    }
                }
            }
        }
        __threadfence();        // the slot is rewritten by the next pair
    }
}

// W[r][c] = adjoint of the argument at point pair (r, c): the adjoint of the double increment (signature_algs.py:26)
//   Gam[r][c] = Lam[r-1][c-1] - Lam[r-1][c] - Lam[r][c-1] + Lam[r][c]   (zero outside the lattice; no differences: Gam = Lam)
// times d kappa / d a.  One thread per point pair; W in the layout of arg.
template <bool RBF>
__global__ void wide_lattice_adjoint_kernel(const WideLatArgs A, double* __restrict__ W) {
    __shared__ double etab[EXP_TAB_N];
    exp_tab_fill(etab, threadIdx.x, blockDim.x);
    __syncthreads();
    const WideKap K = wide_kap(A.kind, etab);
    const int dr = A.difference ? 1 : 0, R1 = A.L1 - dr, R2 = A.L2 - dr;
    const int64_t cells = int64_t(A.L1) * A.L2, total = A.P * cells;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t pp = idx / cells, pg = A.p0 + pp;
        const int r = int((idx % cells) / A.L2), c = int(idx % A.L2);
        const double* lm = A.lam + pp * int64_t(R1) * R2;
        auto at = [&](int a, int b) -> double { return (a >= 0 && a < R1 && b >= 0 && b < R2) ? lm[int64_t(a) * R2 + b] : 0.0; };
        const double gam = dr ? at(r - 1, c - 1) - at(r - 1, c) - at(r, c - 1) + at(r, c) : at(r, c);
        const int64_t off = (pg / A.N2) * A.si + (pg % A.N2) * A.sj + int64_t(r) * A.ld + c;
        double k, dk;
        wide_kappa_grad<RBF>(K, A.arg[off], k, dk);
        W[off] = gam * dk;
    }
}

// The symmetric Gram's upstream gradient folded onto the pairs i <= j (the levels are symmetric functions of the two sequences):
// Gs[m][i][j] = G[m][i][j] + G[m][j][i] (j > i),  G[m][i][i] (j == i),  0 (j < i).
__global__ void wide_sym_upstream_kernel(const double* __restrict__ G, int64_t N, int M1, double* __restrict__ Gs) {
    const int64_t total = int64_t(M1) * N * N;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t j = idx % N, i = (idx / N) % N, m = idx / (N * N);
        Gs[idx] = j > i ? G[idx] + G[(m * N + j) * N + i] : (j == i ? G[idx] : 0.0);
    }
}

// dM[pair][a][b] (signature_algs.py:26 / :56) from the argument lattices, for the sweeps that read it from memory (the higher-order reverse pass,
// grad_wave_ho_kernel.hpp).  One thread per lattice cell.
template <bool RBF>
__global__ void wide_lattice_dm_kernel(const WideLatArgs A, double* __restrict__ dM) {
    __shared__ double etab[EXP_TAB_N];
    exp_tab_fill(etab, threadIdx.x, blockDim.x);
    __syncthreads();
    const WideKap K = wide_kap(A.kind, etab);
    const int dr = A.difference ? 1 : 0, R1 = A.L1 - dr, R2 = A.L2 - dr;
    const int64_t cells = int64_t(R1) * R2, total = A.P * cells;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t pp = idx / cells, pg = A.p0 + pp;
        const int a = int((idx % cells) / R2), b = int(idx % R2);
        const double* base = A.arg + (pg / A.N2) * A.si + (pg % A.N2) * A.sj + int64_t(a) * A.ld + b;
        double v = wide_kappa<RBF>(K, base[0]);
        if (dr) v = (wide_kappa<RBF>(K, base[A.ld + 1]) - wide_kappa<RBF>(K, base[A.ld])) - (wide_kappa<RBF>(K, base[1]) - v);
        dM[idx] = v;
    }
}

// g[r][f] = left-form chain rule of (gl, vl) [+ right-form chain rule of (gr, vr) where given: one array on both sides of the lattices]
__global__ void wide_unaug_pair_kernel(const double* __restrict__ gl, const double* __restrict__ vl, const double* __restrict__ gr, const double* __restrict__ vr,
                                       int64_t rows, int d, double* __restrict__ g) {
    const int DA = d + 2;
    const int64_t total = rows * d;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int f = int(idx % d);
        const int64_t r = idx / d;
        double v = fma(-gl[r * DA + d], vl[r * DA + f], gl[r * DA + f]);
        if (gr) v += fma(-gr[r * DA + d + 1], vr[r * DA + f], gr[r * DA + f]);
        g[idx] = v;
    }
}

// =====================================================================================================================================
// Inducing tensors vs inducing tensors (kernels.py:263-283 + signature_algs.py:76-99) from the argument blocks of every component:
// arg (lt, E * Tpad, E * Tpad), block k = left-form rows of component k times its right-form rows; row / column (e, t) at e * Tpad + t.
// One thread per (t, t'), lanes along t' (contiguous).
struct WideTensArgs {
    const double* arg;
    int64_t Tpad, Tn;
    int32_t M, E, kind, sum_levels;
    const double* w;        // (M+1) weights or NULL
    double* out;            // forward: (T, T) weighted level sum or (M+1, T, T)
    const double* G;        // reverse: (M+1, T, T)
    double* W;              // reverse: adjoint of arg, same layout
};

// value of component k at (t, t'): kappa, or the four-term difference of its two points on both sides (kernels.py:276-277)
template <bool RBF>
__device__ __forceinline__ double wide_tens_val(const double* __restrict__ blk, int64_t R, int64_t Tpad, int E, const WideKap& K, int64_t t, int64_t tp) {
    if (E == 2)
        return wide_kappa<RBF>(K, blk[(Tpad + t) * R + Tpad + tp]) + wide_kappa<RBF>(K, blk[t * R + tp]) - wide_kappa<RBF>(K, blk[(Tpad + t) * R + tp]) -
               wide_kappa<RBF>(K, blk[t * R + Tpad + tp]);
    return wide_kappa<RBF>(K, blk[t * R + tp]);
}

template <bool RBF>
__global__ void __launch_bounds__(64) wide_tens_fwd_kernel(const WideTensArgs A) {
    __shared__ double etab[EXP_TAB_N];
    exp_tab_fill(etab, threadIdx.x, 64);
    __syncthreads();
    const WideKap K = wide_kap(A.kind, etab);
    const int64_t tp = int64_t(blockIdx.x) * 64 + threadIdx.x, R = int64_t(A.E) * A.Tpad;
    if (tp >= A.Tn) return;
    for (int64_t t = blockIdx.y; t < A.Tn; t += gridDim.y) {
        double acc = A.w ? A.w[0] : 1.0;                                   // level 0 == 1 (signature_algs.py:88)
        if (!A.sum_levels) A.out[t * A.Tn + tp] = acc;
        int k = 0;
        for (int i = 1; i <= A.M; ++i) {
            double prod = 1.0;
            for (int j = 0; j < i; ++j, ++k) prod *= wide_tens_val<RBF>(A.arg + int64_t(k) * R * R, R, A.Tpad, A.E, K, t, tp);      // :91-97
            const double f = A.w ? A.w[i] : 1.0;
            if (A.sum_levels) acc = fma(prod, f, acc);
            else A.out[(int64_t(i) * A.Tn + t) * A.Tn + tp] = prod * f;
        }
        if (A.sum_levels) A.out[t * A.Tn + tp] = acc;
    }
}

// grid (Tpad / 64, Tpad rows (grid-stride)): every entry of W is written (zeros at padded tensors)
template <bool RBF>
__global__ void __launch_bounds__(64) wide_tens_bwd_kernel(const WideTensArgs A) {
    __shared__ double etab[EXP_TAB_N];
    exp_tab_fill(etab, threadIdx.x, 64);
    __syncthreads();
    const WideKap K = wide_kap(A.kind, etab);
    const int64_t tp = int64_t(blockIdx.x) * 64 + threadIdx.x, R = int64_t(A.E) * A.Tpad;
    for (int64_t t = blockIdx.y; t < A.Tpad; t += gridDim.y) {
        const bool valid = t < A.Tn && tp < A.Tn;
        int k0 = 0;
        for (int i = 1; i <= A.M; ++i) {
            const double c = valid ? A.G[(int64_t(i) * A.Tn + t) * A.Tn + tp] : 0.0;
            double v[WIDE_MAX_LEVELS];
#pragma unroll
            for (int j = 0; j < WIDE_MAX_LEVELS; ++j) v[j] = j < i ? wide_tens_val<RBF>(A.arg + int64_t(k0 + j) * R * R, R, A.Tpad, A.E, K, t, tp) : 1.0;
#pragma unroll
            for (int j = 0; j < WIDE_MAX_LEVELS; ++j) {
                if (j >= i) continue;
                double g = c;                                              // dL/dval_j = c * prod of the other components
#pragma unroll
                for (int q = 0; q < WIDE_MAX_LEVELS; ++q)
                    if (q != j && q < i) g *= v[q];
                const double* blk = A.arg + int64_t(k0 + j) * R * R;
                double* wb = A.W + int64_t(k0 + j) * R * R;
                double kk, dk;
                if (A.E == 2) {
                    wide_kappa_grad<RBF>(K, blk[(A.Tpad + t) * R + A.Tpad + tp], kk, dk); wb[(A.Tpad + t) * R + A.Tpad + tp] = g * dk;
                    wide_kappa_grad<RBF>(K, blk[t * R + tp], kk, dk);                     wb[t * R + tp] = g * dk;
                    wide_kappa_grad<RBF>(K, blk[(A.Tpad + t) * R + tp], kk, dk);          wb[(A.Tpad + t) * R + tp] = -g * dk;
                    wide_kappa_grad<RBF>(K, blk[t * R + A.Tpad + tp], kk, dk);            wb[t * R + A.Tpad + tp] = -g * dk;
                } else {
                    wide_kappa_grad<RBF>(K, blk[t * R + tp], kk, dk);
                    wb[t * R + tp] = g * dk;
                }
            }
            k0 += i;
        }
    }
}

// gZ in the caller's (lt, T, E, d) order from the adjoints of the left-form and the right-form augmented rows (rows (k * E + e) * Tpad + t)
__global__ void wide_unaug_tens_kernel(const double* __restrict__ gl, const double* __restrict__ vl, const double* __restrict__ gr, const double* __restrict__ vr,
                                       int64_t rows_out, int d, int64_t Tn, int64_t Tpad, int E, double* __restrict__ g) {
    const int DA = d + 2;
    const int64_t total = rows_out * d;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int f = int(idx % d);
        const int64_t ro = idx / d;
        const int e = int(ro % E);
        const int64_t t = (ro / E) % Tn;
        const int k = int(ro / (int64_t(E) * Tn));
        const int64_t r = (int64_t(k) * E + e) * Tpad + t;
        g[idx] = fma(-gl[r * DA + d], vl[r * DA + f], gl[r * DA + f]) + fma(-gr[r * DA + d + 1], vr[r * DA + f], gr[r * DA + f]);
    }
}

// =====================================================================================================================================
// The two contractions of the reverse pass in ONE pass over the adjoint array W (rows = (sequence, time), CW columns), for narrow augmented rows
// (DA <= DAP = 16 / 32):   gZA[c][f] = sum_r W[r][c] XA[r][f]   and   gXA[r][f] = sum_c W[r][c] ZA[c][f].
// rocBLAS runs these as 12 x 10,240-output products over K = 25,000 on 80 macro-tiles (1.1 + 0.7 ms per GB of W at NetFlow's shape: more than every
// hand-written kernel of the step together); here a wavefront owns a strip of 64 rows and walks the columns in tiles of 64: the tile goes to LDS once,
// phase 1 (lane = column) adds its 64 rows into the strip's partial sums of gZA -- XA rows are wavefront-uniform: scalar operands --, phase 2
// (lane = row) adds its 64 columns into gXA, which is complete when the strip ends.  The partial sums (strips, CW, DA) are summed in a fixed order by
// wide_contract_reduce_kernel (deterministic; a quarter of W's bytes at DAP = 16).
struct WideContractArgs {
    const double* W;        // (R, CW)
    const double* XA;       // (R, DA)
    const double* ZA;       // (CW, DA)
    int64_t R, CW;
    int32_t DA, groups;     // column tiles are dealt to `groups` workgroups per strip (grid y): enough wavefronts to fill the chip
    double* gXA_part;       // (groups, R, DAP)
    double* part;           // (strips, CW, DAP)
};

// grid (strips, groups), one wavefront each.  Both products of a tile on the float64 matrix cores (v_mfma_f64_16x16x4; operand layout as
// lowrank_kernels.hpp: A: lane l holds A[l & 15][l >> 4], B: B[l >> 4][l & 15], results D[(l >> 4) + 4 r][l & 15]): per 64 x 64 tile and 16 columns of
// the augmented rows 64 + 64 MFMAs against 80 + 80 LDS reads -- the first form (one broadcast LDS read per FMA) ran at rocBLAS's 3.5 ms.
typedef double wide_f64x4 __attribute__((ext_vector_type(4)));

template <int DAP>
__global__ void __launch_bounds__(64) wide_contract_kernel(const WideContractArgs A) {
    constexpr int TS = 65, NB = DAP / 16;
    __shared__ double tile[64 * TS];
    __shared__ double xs[64 * DAP], zs[64 * DAP];            // the strip's rows of XA, the tile's rows of ZA: zero beyond DA
    const int lane = threadIdx.x, DA = A.DA, li = lane & 15, lk = lane >> 4;
    const int64_t ntiles = (A.CW + 63) / 64;
    for (int64_t strip = blockIdx.x; strip * 64 < A.R; strip += gridDim.x) {
        const int64_t r0 = strip * 64;
        const int nr = int(A.R - r0 < 64 ? A.R - r0 : 64);
        __syncthreads();
        for (int e = lane; e < 64 * DAP; e += 64) {
            const int r = e / DAP, f = e % DAP;
            xs[e] = (r < nr && f < DA) ? A.XA[(r0 + r) * DA + f] : 0.0;
        }
        wide_f64x4 acc2[4][NB];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc2[m][n] = wide_f64x4{0.0, 0.0, 0.0, 0.0};
        // a tile's 64 rows of W are requested a whole tile ahead (64 registers): a wavefront has nothing else to hide that latency behind
        double wn[64];
        auto request = [&](int64_t ti) {
            const int64_t c = ti * 64 + lane;
            const double* __restrict__ src = A.W + r0 * A.CW + (c < A.CW ? c : 0);
#pragma unroll
            for (int r = 0; r < 64; ++r) wn[r] = src[int64_t(r < nr ? r : 0) * A.CW];
        };
        if (int64_t(blockIdx.y) < ntiles) request(blockIdx.y);
        for (int64_t ti = blockIdx.y; ti < ntiles; ti += A.groups) {
            const int64_t c0 = ti * 64;
            const bool cok = c0 + lane < A.CW;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 64; ++r) tile[r * TS + lane] = (r < nr && cok) ? wn[r] : 0.0;
            if (ti + A.groups < ntiles) request(ti + A.groups);
            for (int e = lane; e < 64 * DAP; e += 64) {
                const int cc = e / DAP, f = e % DAP;
                zs[e] = (c0 + cc < A.CW && f < DA) ? A.ZA[(c0 + cc) * DA + f] : 0.0;
            }
            __syncthreads();
            // phase 1: D1[column][f] = sum over the strip's rows of W[row][column] XA[row][f]
            wide_f64x4 acc1[4][NB];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc1[m][n] = wide_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
            for (int k0 = 0; k0 < 64; k0 += 4) {
                double av[4], bv[NB];
#pragma unroll
                for (int m = 0; m < 4; ++m) av[m] = tile[(k0 + lk) * TS + m * 16 + li];
#pragma unroll
                for (int n = 0; n < NB; ++n) bv[n] = xs[(k0 + lk) * DAP + n * 16 + li];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n) acc1[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[n], acc1[m][n], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t col = c0 + m * 16 + lk + 4 * r;
                    if (col < A.CW) {
#pragma unroll
                        for (int n = 0; n < NB; ++n) A.part[(strip * A.CW + col) * DAP + n * 16 + li] = acc1[m][n][r];
                    }
                }
            // phase 2: D2[row][f] += sum over the tile's columns of W[row][column] ZA[column][f]
#pragma unroll 4
            for (int k0 = 0; k0 < 64; k0 += 4) {
                double av[4], bv[NB];
#pragma unroll
                for (int m = 0; m < 4; ++m) av[m] = tile[(m * 16 + li) * TS + k0 + lk];
#pragma unroll
                for (int n = 0; n < NB; ++n) bv[n] = zs[(k0 + lk) * DAP + n * 16 + li];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n) acc2[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[n], acc2[m][n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m * 16 + lk + 4 * r;
                if (row < nr) {
#pragma unroll
                    for (int n = 0; n < NB; ++n) A.gXA_part[(int64_t(blockIdx.y) * A.R + r0 + row) * DAP + n * 16 + li] = acc2[m][n][r];
                }
            }
    }
}

// Second form (round 6, DAP = 16): the same tile walk with PART tiles of HR = 32 / 16 rows in LDS (16.6 / 8.3 KB per wavefront instead of 49.6: two wavefronts per SIMD where the
// first form fits three per CU -- its matrix cores idle while its only wavefront loads and stages), and the MFMAs' second operands in REGISTERS: a lane's 16
// entries of the strip's XA rows are loaded once per strip, its 16 entries of the tile's ZA rows once per tile.  MFMA count and partial-sum layout unchanged.
template <int HR>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) wide_contract16_kernel(const WideContractArgs A) {
    constexpr int TS = 65, NH = 64 / HR, Q1 = HR / 4, MB = HR / 16;     // part tiles of HR rows: NH per tile, Q1 reduction steps of phase 1, MB row blocks of phase 2
    __shared__ double tile[HR * TS];
    const int lane = threadIdx.x, DA = A.DA, li = lane & 15, lk = lane >> 4;
    const int64_t ntiles = (A.CW + 63) / 64;
    const bool fok = li < DA;
    for (int64_t strip = blockIdx.x; strip * 64 < A.R; strip += gridDim.x) {
        const int64_t r0 = strip * 64;
        const int nr = int(A.R - r0 < 64 ? A.R - r0 : 64);
        // this lane's entries of the strip's rows of XA as MFMA B operands: step s of half h covers rows 32 h + 4 s + lk
        double bx[NH][Q1];
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int q = 0; q < Q1; ++q) {
                const int row = HR * h + 4 * q + lk;
                bx[h][q] = (row < nr && fok) ? A.XA[(r0 + row) * DA + li] : 0.0;
            }
        wide_f64x4 acc2[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc2[m] = wide_f64x4{0.0, 0.0, 0.0, 0.0};
        double wn[HR];
        auto request = [&](int64_t ti, int h) {
            const int64_t c = ti * 64 + lane;
            const int hb = HR * h < nr ? HR * h : 0;             // (a part tile wholly beyond the strip's rows: any rows inside it, masked at the LDS write)
            const double* __restrict__ src = A.W + (r0 + hb) * A.CW + (c < A.CW ? c : 0);
#pragma unroll
            for (int r = 0; r < HR; ++r) wn[r] = src[int64_t(hb + r < nr ? r : 0) * A.CW];
        };
        if (int64_t(blockIdx.y) < ntiles) request(blockIdx.y, 0);
        for (int64_t ti = blockIdx.y; ti < ntiles; ti += A.groups) {
            const int64_t c0 = ti * 64;
            const bool cok = c0 + lane < A.CW;
            double bz[16];                                   // this lane's entries of the tile's rows of ZA: step s covers columns 4 s + lk
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t col = c0 + 4 * q + lk;
                bz[q] = (col < A.CW && fok) ? A.ZA[col * DA + li] : 0.0;
            }
            wide_f64x4 acc1[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc1[m] = wide_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < HR; ++r) tile[r * TS + lane] = (HR * h + r < nr && cok) ? wn[r] : 0.0;
                if (h + 1 < NH) request(ti, h + 1);
                else if (ti + A.groups < ntiles) request(ti + A.groups, 0);
                __syncthreads();
                // phase 1: D1[column][f] += sum over the half's rows of W[row][column] XA[row][f]
#pragma unroll
                for (int q = 0; q < Q1; ++q) {
                    double av[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) av[m] = tile[(4 * q + lk) * TS + m * 16 + li];
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc1[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bx[h][q], acc1[m], 0, 0, 0);
                }
                // phase 2: D2[row][f] += sum over the tile's columns of W[row][column] ZA[column][f], the half's 32 rows
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    double av[MB];
#pragma unroll
                    for (int m = 0; m < MB; ++m) av[m] = tile[(m * 16 + li) * TS + 4 * q + lk];
#pragma unroll
                    for (int m = 0; m < MB; ++m) acc2[MB * h + m] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bz[q], acc2[MB * h + m], 0, 0, 0);
                }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t col = c0 + m * 16 + lk + 4 * r;
                    if (col < A.CW) A.part[(strip * A.CW + col) * 16 + li] = acc1[m][r];
                }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m * 16 + lk + 4 * r;
                if (row < nr) A.gXA_part[(int64_t(blockIdx.y) * A.R + r0 + row) * 16 + li] = acc2[m][r];
            }
    }
}

// dst[e][f] = (acc ? dst : 0) + sum over k of part[k][e][f], f < DA of DAP (a fixed order: deterministic)
__global__ void wide_contract_reduce_kernel(const double* __restrict__ part, int64_t nparts, int64_t rows, int DA, int DAP, int acc, double* __restrict__ dst) {
    const int64_t n = rows * DA;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < n; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t e = idx / DA;
        const int f = int(idx % DA);
        double s = acc ? dst[idx] : 0.0;
        for (int64_t k = 0; k < nparts; ++k) s += part[(k * rows + e) * DAP + f];
        dst[idx] = s;
    }
}

}  // namespace gpsig
