// lr_grad_kernel.hpp -- reverse pass of the low-rank feature map of a batch of sequences (lr_fused_kernel.hpp) in ONE kernel (round 4).
//
// The reference trains low-rank mode through TensorFlow's autodiff of Nystrom_map (gpsig/low_rank_calculations.py:26-61), the running
// sums and the sparse projections of signature_kern_first_order_lr_feature (gpsig/signature_algs.py:162-192); there is no gradient code to
// restate.  Round 3 ran that reverse pass as torch ops (gather x gather x value, index_add: gpsig_amd/autodiff.py), which moves every
// (N, L, nnz) product through HBM: 9.1 s for the SVGP covariances of BASELINE configs[2] against 14 ms in exact mode.  Here a workgroup
// owns one sequence at a time and keeps four (width, L) arrays in LDS, like the forward kernel keeps three:
//
//   forward again   x, kxs = kappa(x, S), feat = kxs Wh, U = time difference; E_2 = excumsum_t(U); E_{i+1} = excumsum_t(sketch_i(U, E_i))
//                   -- the E_i go to a per-WORKGROUP scratch (a few tens of KB each: L2-resident), nothing per sequence is stored in HBM
//   levels M .. 2   dP_M[t] = g_M (Phi_M = sum_t P_M);  dU[i1] += sum_e val E_i[i2] dP_i[j]   (the sketch's entries grouped by i1)
//                   dE_i[i2] = sum_e val U[i1] dP_i[j]  (grouped by i2);  dP_{i-1}[t] = g_{i-1} + sum_{t' > t} dE_i[t']
//                   -- both are GATHERS over transposed copies of the sketch (built once per draw on the host), the same loop as the
//                   forward sketch: entries through the scalar unit, operands full-width LDS reads, no atomics, deterministic
//   level 1         dU += dP_1;  dfeat = the time difference's adjoint;  dkxs = dfeat Wh^T;  dWh += kxs^T dfeat
//   base kernel     dx[t] = sum_i dkxs[t][i] d kappa(x_t, S_i) / dx,  dS_i += sum_t dkxs[t][i] d kappa / dS_i  (base_eval_grad, grad_core.hpp)
//
// dWh, dS and the base kernel's own parameter are summed over the sequences a workgroup processes in registers and leave as one partial
// per workgroup (added up by lr_grad_reduce_kernel in a fixed order).  Layout and lane mappings as in lr_fused_kernel.hpp: arrays
// [column][time] with an odd row stride, lane = time for everything elementwise in time, thread = column for the running sums.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aux_kernels.hpp"
#include "grad_core.hpp"
#include "lr_fused_args.hpp"

namespace gpsig {

struct LrGradSketch {
    const int32_t* colptr; const LrEntry* ent;        // by output column j: (val, i1, i2)                      P[j]   = sum val U[i1] E[i2]
    const int32_t* ptr1; const LrEntry* ent1;         // by i1: (val, i2, j) in the entry's (i1, i2) fields        dU[i1] += sum val E[i2] dP[j]
    const int32_t* ptr2; const LrEntry* ent2;         // by i2: (val, i1, j)                                        dE[i2]  = sum val U[i1] dP[j]
};

struct LrGradArgs {
    const double* X; int64_t N; int L, d;              // sequences as given (N, L, d): the level primitives take scaled inputs
    const double* S;                                   // landmarks (c, d)
    const double* Wh;                                  // whitening (c, c): feat[j] = sum_i kxs[i] Wh[i][j]
    int c, r, M, difference, kind;
    double p0, p1;
    LrGradSketch sk[LR_FUSED_MAX_SKETCHES];
    const double* dPhi; int F;                         // upstream (N, F)
    double* gX;                                        // (N, L, d)
    double* part;                                      // per-workgroup partial sums [grid][c d + c c + 1]: dS, dWh, d base parameter
    double* escr; int64_t escr_stride;                 // per-workgroup scratch for E_2 .. E_M (doubles per workgroup)
    int lp, rows_b;
};

constexpr int LR_GRAD_THREADS = 512;                   // (the smaller of the two workgroup sizes built: what the per-thread tables are sized for)
constexpr int LR_GRAD_KW = 8, LR_GRAD_KS = 8;          // (i, j) pairs of dWh and (i, f) pairs of dS per thread: c <= 64, c d <= 4096

inline size_t lr_grad_lds_bytes(int c, int r, int d, int L, int pad = 1) {
    const int lp = lr_fused_stride(L, pad);
    int kb = c > r ? c : r;
    if (d > kb) kb = d;
    if (kb < 16) kb = 16;                              // (a row per wavefront for the per-wave partial sums: up to 1024 threads)
    return sizeof(double) * size_t(lp) * 4 * size_t(kb);
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void lr_seq_features_grad_kernel(LrGradArgs A) {
    constexpr int NW = THREADS / 64, UNROLL = 8;
    extern __shared__ double lrg_lds[];
    const int lp = A.lp, c = A.c, r = A.r, L = A.L, d = A.d, M = A.M;
    double* const B0 = lrg_lds;                                 // U; later kxs; later per-wave partial sums
    double* const B1 = B0 + size_t(A.rows_b) * lp;              // x; dU; x again
    double* BX = B1 + size_t(A.rows_b) * lp;
    double* BY = BX + size_t(A.rows_b) * lp;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = A.difference ? L - 1 : L;                     // time steps of U
    const int nchunk = (L + 63) / 64;
    double* const escr = A.escr + int64_t(blockIdx.x) * A.escr_stride;
    const lr_const_ptr<double> Sg = lr_as_const(A.S);
    const lr_const_ptr<double> Whg = lr_as_const(A.Wh);

    // out[row][t] (+)= sum_e val * a[ia][t] * b[ib][t] over the entries of `row`: the forward sketch's loop
    auto apply = [&](const int32_t* ptr_, const LrEntry* ent_, int nrows, const double* a, const double* b, double* out, bool accumulate) {
        const lr_const_ptr<int32_t> ptr = lr_as_const(ptr_);
        const lr_const_ptr<LrEntry> ent = lr_as_const(ent_);
        for (int row = wave; row < nrows; row += NW) {
            const int e0 = ptr[row], e1 = ptr[row + 1];
            for (int ch = 0; ch < nchunk; ++ch) {
                const int t = ch * 64 + lane;
                const int tt = t < l ? t : 0;                   // idle lanes read a valid address
                double acc = 0.0;
#pragma unroll UNROLL
                for (int e = e0; e < e1; ++e) {
                    const double val = ent[e].val;
                    const int ia = ent[e].i1, ib = ent[e].i2;
                    acc = fma(val * a[ia * lp + tt], b[ib * lp + tt], acc);
                }
                if (t < l) out[row * lp + t] = accumulate ? out[row * lp + t] + acc : acc;
            }
        }
    };
    // x[f][t] and kxs[i][t] = kappa(x_t, S_i) of the current sequence
    auto load_x = [&](const double* Xn, double* xb) {
        for (int q = threadIdx.x; q < L * d; q += THREADS) {
            const int t = q / d, f = q - t * d;
            xb[f * lp + t] = Xn[q];
        }
    };
    auto cross = [&](const double* xb, double* kb) {
        for (int ch = 0; ch < nchunk; ++ch) {
            const int t = ch * 64 + lane;
            if (t < L) {
                double xs = 0.0;
                for (int f = 0; f < d; ++f) { const double x = xb[f * lp + t]; xs = fma(x, x, xs); }
                for (int i = wave; i < c; i += NW) {
                    double ip = 0.0, ss = 0.0;
                    for (int f = 0; f < d; ++f) {
                        const double y = Sg[size_t(i) * d + f];
                        ip = fma(xb[f * lp + t], y, ip);
                        ss = fma(y, y, ss);
                    }
                    kb[i * lp + t] = base_eval<double>(A.kind, ip, xs, ss, A.p0, A.p1);
                }
            }
        }
    };

    double accW[LR_GRAD_KW], accS[LR_GRAD_KS], accP = 0.0;      // this workgroup's sums over its sequences: dWh, dS, d base parameter
#pragma unroll
    for (int k = 0; k < LR_GRAD_KW; ++k) accW[k] = 0.0;
#pragma unroll
    for (int k = 0; k < LR_GRAD_KS; ++k) accS[k] = 0.0;

    for (int64_t n = blockIdx.x; n < A.N; n += gridDim.x) {
        const double* Xn = A.X + n * int64_t(L) * d;
        const double* g = A.dPhi + n * int64_t(A.F);
        __syncthreads();
        // ---- forward again: x -> B1, kxs -> BX, feat -> BY, U -> B0
        load_x(Xn, B1);
        __syncthreads();
        cross(B1, BX);
        __syncthreads();
        for (int ch = 0; ch < nchunk; ++ch) {
            const int t = ch * 64 + lane;
            if (t < L) {
                for (int j = wave; j < c; j += NW) {
                    double acc = 0.0;
#pragma unroll 4
                    for (int i = 0; i < c; ++i) acc = fma(BX[i * lp + t], Whg[size_t(i) * c + j], acc);
                    BY[j * lp + t] = acc;
                }
            }
        }
        __syncthreads();
        for (int ch = 0; ch < nchunk; ++ch) {
            const int t = ch * 64 + lane;
            if (t < l) {
                for (int j = wave; j < c; j += NW) {
                    const double f0 = BY[j * lp + t];
                    B0[j * lp + t] = A.difference ? BY[j * lp + t + 1] - f0 : f0;
                }
            }
        }
        __syncthreads();
        // E_2 = excumsum_t(U) -> BX and the scratch; then E_{i+1} = excumsum_t(sketch_i(U, E_i)), i = 2 .. M-1
        double* cur = BX;
        double* nxt = BY;
        if (M >= 2) {
            for (int j = threadIdx.x; j < c; j += THREADS) {
                double run = 0.0;
                const double* u = B0 + size_t(j) * lp;
                double* e = cur + size_t(j) * lp;
                double* es = escr + size_t(j) * l;
                for (int t = 0; t < l; ++t) { const double v = u[t]; e[t] = run; es[t] = run; run += v; }
            }
            __syncthreads();
            int64_t eo = int64_t(c) * l;
            for (int lev = 2; lev < M; ++lev) {
                apply(A.sk[lev - 2].colptr, A.sk[lev - 2].ent, r, B0, cur, nxt, false);
                __syncthreads();
                for (int j = threadIdx.x; j < r; j += THREADS) {
                    double run = 0.0;
                    double* e = nxt + size_t(j) * lp;
                    double* es = escr + eo + size_t(j) * l;
                    for (int t = 0; t < l; ++t) { const double v = e[t]; e[t] = run; es[t] = run; run += v; }
                }
                __syncthreads();
                eo += int64_t(r) * l;
                double* tmp = cur; cur = nxt; nxt = tmp;
            }
        }
        // ---- backward through the levels.  dU -> B1 (x is read again from memory later)
        for (int q = threadIdx.x; q < c * lp; q += THREADS) B1[q] = 0.0;
        double* Y = nxt;                            // dP of the level being processed
        double* Xb = cur;                           // E of that level, then dE
        if (M >= 2) {
            const double* gM = g + 1 + c + (M - 2) * r;
            for (int ch = 0; ch < nchunk; ++ch) {
                const int t = ch * 64 + lane;
                if (t < l)
                    for (int j = wave; j < r; j += NW) Y[j * lp + t] = gM[j];
            }
        }
        __syncthreads();
        for (int lev = M; lev >= 2; --lev) {
            const int w = lev == 2 ? c : r;         // width of E_lev
            // E_lev from the scratch (the forward pass left E_{M} in `cur` already)
            int64_t eo = 0;
            for (int k = 2; k < lev; ++k) eo += int64_t(k == 2 ? c : r) * l;
            if (lev != M || true) {
                for (int q = threadIdx.x; q < w * l; q += THREADS) {
                    const int j = q / l, t = q - j * l;
                    Xb[j * lp + t] = escr[eo + q];
                }
            }
            __syncthreads();
            const LrGradSketch sk = A.sk[lev - 2];
            apply(sk.ptr1, sk.ent1, c, Xb, Y, B1, true);              // dU[i1] += val E[i2] dP[j]
            __syncthreads();
            apply(sk.ptr2, sk.ent2, w, B0, Y, Xb, false);             // dE[i2]  = val U[i1] dP[j]   (E is no longer needed)
            __syncthreads();
            // dP_{lev-1}[t] = g_{lev-1} + sum_{t' > t} dE[t'], in place
            const double* gl = lev - 1 == 1 ? g + 1 : g + 1 + c + (lev - 3) * r;
            for (int j = threadIdx.x; j < w; j += THREADS) {
                double run = gl[j];
                double* e = Xb + size_t(j) * lp;
                for (int t = l - 1; t >= 0; --t) { const double v = e[t]; e[t] = run; run += v; }
            }
            __syncthreads();
            double* tmp = Xb; Xb = Y; Y = tmp;      // Y: dP_{lev-1}
        }
        // level 1: Phi_1 = sum_t U (M == 1: that is all there is)
        for (int ch = 0; ch < nchunk; ++ch) {
            const int t = ch * 64 + lane;
            if (t < l)
                for (int j = wave; j < c; j += NW) B1[j * lp + t] += M >= 2 ? Y[j * lp + t] : g[1 + j];
        }
        __syncthreads();
        // dfeat[j][t] -> Xb: the adjoint of the time difference (signature_algs.py:180)
        for (int ch = 0; ch < nchunk; ++ch) {
            const int t = ch * 64 + lane;
            if (t < L)
                for (int j = wave; j < c; j += NW)
                    Xb[j * lp + t] = A.difference ? (t >= 1 ? B1[j * lp + t - 1] : 0.0) - (t < l ? B1[j * lp + t] : 0.0) : B1[j * lp + t];
        }
        __syncthreads();
        // x -> B1 and kxs -> B0 once more (U and dU are done with)
        load_x(Xn, B1);
        __syncthreads();
        cross(B1, B0);
        __syncthreads();
        // dWh[i][j] += sum_t kxs[i][t] dfeat[j][t]
#pragma unroll
        for (int k = 0; k < LR_GRAD_KW; ++k) {
            const int q = k * THREADS + threadIdx.x;
            if (q < c * c) {
                const int i = q / c, j = q - i * c;
                double acc = 0.0;
                for (int t = 0; t < L; ++t) acc = fma(B0[i * lp + t], Xb[j * lp + t], acc);
                accW[k] += acc;
            }
        }
        // dkxs[i][t] = sum_j dfeat[j][t] Wh[i][j] -> Y
        for (int ch = 0; ch < nchunk; ++ch) {
            const int t = ch * 64 + lane;
            if (t < L) {
                for (int i = wave; i < c; i += NW) {
                    double acc = 0.0;
#pragma unroll 4
                    for (int j = 0; j < c; ++j) acc = fma(Xb[j * lp + t], Whg[size_t(i) * c + j], acc);
                    Y[i * lp + t] = acc;
                }
            }
        }
        __syncthreads();
        // through the base kernel: d kappa / dx = wy S_i + wx x,  d kappa / dS_i = wy x + ws S_i  (BaseGrad of grad_core.hpp)
        //   Y[i][t] <- dkxs wy,  Xb[i][t] <- dkxs ws,  B0[wave][t] <- this wave's share of sum_i dkxs wx
        for (int ch = 0; ch < nchunk; ++ch) {
            const int t = ch * 64 + lane;
            double ax = 0.0;
            if (t < L) {
                double xs = 0.0;
                for (int f = 0; f < d; ++f) { const double x = B1[f * lp + t]; xs = fma(x, x, xs); }
                for (int i = wave; i < c; i += NW) {
                    double ip = 0.0, ss = 0.0;
                    for (int f = 0; f < d; ++f) {
                        const double y = Sg[size_t(i) * d + f];
                        ip = fma(B1[f * lp + t], y, ip);
                        ss = fma(y, y, ss);
                    }
                    const BaseGrad bg = base_eval_grad(A.kind, ip, xs, ss, A.p0, A.p1);
                    const double dk = Y[i * lp + t];
                    Y[i * lp + t] = dk * (bg.cy - bg.cd);
                    Xb[i * lp + t] = dk * (bg.cx2 + bg.cd);
                    ax = fma(dk, bg.cx + bg.cd, ax);
                    accP = fma(dk, bg.dp0, accP);
                }
            }
            __syncthreads();                         // (kxs in B0 was last read by the dWh sums above, before the previous barrier)
            if (t < L) B0[wave * lp + t] = ax;
            __syncthreads();
            // dx[t][f] = sum_i Y[i][t] S_i[f] + x[f][t] sum_w B0[w][t]
            if (t < L) {
                double axs = 0.0;
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) axs += B0[w2 * lp + t];
                for (int f = wave; f < d; f += NW) {
                    double acc = axs * B1[f * lp + t];
                    for (int i = 0; i < c; ++i) acc = fma(Y[i * lp + t], Sg[size_t(i) * d + f], acc);
                    A.gX[(n * int64_t(L) + t) * d + f] = acc;
                }
            }
            __syncthreads();
        }
        // dS[i][f] += sum_t Y[i][t] x[f][t] + S_i[f] sum_t Xb[i][t]
#pragma unroll
        for (int k = 0; k < LR_GRAD_KS; ++k) {
            const int q = k * THREADS + threadIdx.x;
            if (q < c * d) {
                const int i = q / d, f = q - i * d;
                double a1 = 0.0, a2 = 0.0;
                for (int t = 0; t < L; ++t) {
                    a1 = fma(Y[i * lp + t], B1[f * lp + t], a1);
                    a2 += Xb[i * lp + t];
                }
                accS[k] += fma(a2, Sg[size_t(i) * d + f], a1);
            }
        }
    }
    // ---- this workgroup's partial sums
    double* part = A.part + int64_t(blockIdx.x) * (int64_t(c) * d + int64_t(c) * c + 1);
#pragma unroll
    for (int k = 0; k < LR_GRAD_KS; ++k) {
        const int q = k * THREADS + threadIdx.x;
        if (q < c * d) part[q] = accS[k];
    }
#pragma unroll
    for (int k = 0; k < LR_GRAD_KW; ++k) {
        const int q = k * THREADS + threadIdx.x;
        if (q < c * c) part[int64_t(c) * d + q] = accW[k];
    }
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) accP += __shfl_xor(accP, o, 64);
    if (lane == 0) lrg_lds[wave] = accP;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w2 = 0; w2 < NW; ++w2) s += lrg_lds[w2];
        part[int64_t(c) * d + int64_t(c) * c] = s;
    }
}

// out[q] = sum over the workgroups' partials, in order
__global__ void lr_grad_reduce_kernel(const double* __restrict__ part, int nparts, int64_t width, double* __restrict__ gS, int64_t nS,
                                      double* __restrict__ gWh, int64_t nW, double* __restrict__ gp) {
    const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q >= width) return;
    double s = 0.0;
    for (int k = 0; k < nparts; ++k) s += part[int64_t(k) * width + q];
    if (q < nS) gS[q] = s;
    else if (q < nS + nW) gWh[q - nS] = s;
    else if (gp) gp[0] = s;
}

}  // namespace gpsig
