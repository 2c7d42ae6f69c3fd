// aux_kernels.hpp -- the small kernels around the pair recursion: input preparation (lengthscale
// division, lags, increments, record layout), per-sequence normalisation factors, and the
// inducing-tensor kernels (tensor-vs-sequence scans, tensor-vs-tensor products).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "seq_core.hpp"

namespace gpsig {

constexpr int MAX_FEATURES = 64;  // d (one lag copy) whose lengthscales travel by value in ScaleParams; wider state spaces read them from
                                  // device memory (ScaleParams::ls_dev).  More than 32 columns after lags run through the any-shape kernels
constexpr int MAX_FEATURES_WIDE = 4096;
constexpr int MAX_LAGS = 8;

// Scaling state of SignatureKernel (gpsig/kernels.py:343-398), by value in kernel arguments.
struct ScaleParams {
    int d_in;        // num_features
    int num_lags;    // p
    int has_ls;
    double inv_unused;
    double ls[MAX_FEATURES];
    const double* ls_dev;   // d_in > MAX_FEATURES with lengthscales: their device copy (NULL otherwise)
    double lags[MAX_LAGS];
    double gamma[MAX_LAGS + 1];
    double jitter;
    __host__ __device__ int d_eff() const { return d_in * (num_lags + 1); }
    __host__ __device__ double lsv(int f) const { return ls_dev ? ls_dev[f] : ls[f]; }
};

// x~[n][t][fe]: observation t of sequence n after add_lags_to_sequences (gpsig/lags.py:41-63), division by
// the lengthscales (kernels.py:357-358) and the lag weights gamma (kernels.py:360-361).  fe = lag * d_in + f.
template <typename T>
__device__ __forceinline__ T scaled_point(const T* __restrict__ Xn, int L, int t, int fe, const ScaleParams& P) {
    const int lag = fe / P.d_in, f = fe - lag * P.d_in;
    T v;
    if (lag == 0) {
        v = Xn[int64_t(t) * P.d_in + f];
    } else {
        // lin_interp (gpsig/lags.py:7-38): left = the last grid time not later than the query (+jitter)
        const T denom = T(L - 1);
        const T tq = fmax(T(t) / denom - T(P.lags[lag - 1]), T(0));     // lags.py:56-57
        int left = 0;
        for (int i = L - 1; i >= 0; --i)
            if (!(T(i) / denom - tq > T(P.jitter))) { left = i; break; }  // lags.py:20-22
        const int right = left + 1 < L ? left + 1 : L - 1;              // lags.py:23 (never out of range for lags > 0)
        const T xl = Xn[int64_t(left) * P.d_in + f], xr = Xn[int64_t(right) * P.d_in + f];
        const T tl = T(left) / denom, tr = T(right) / denom;
        v = xl + (tq - tl) * (xr - xl) / (tr - tl);                     // lags.py:33
    }
    if (P.has_ls) v = v / T(P.lsv(f));
    if (P.num_lags > 0) v = v * T(P.gamma[lag]);
    return v;
}

// Records for the seq-gram kernel (layout: seq_configs.hpp / SeqGeom).  One thread per (n, row, fe).
// `out` must have been zero-filled (padding columns and the record tail stay zero).
// norm_col >= 0 (point rows, float64 RBF kernels with the table-driven exp): the values are multiplied by `pre`
// (EXP_PRESCALE) and column norm_col of every row receives -|row|^2 / 2 (seq_step_rbf_prescaled in seq_core.hpp).
template <typename T>
__global__ void prep_seq_records_kernel(const T* __restrict__ X, int64_t N, int L, ScaleParams P, int mode,
                                        int difference, int rows, int RS, int64_t rec_elems, T* __restrict__ out,
                                        T pre = T(1), int norm_col = -1) {
    const int d_eff = P.d_eff();
    const int64_t total = N * rows * d_eff;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int fe = int(idx % d_eff);
        const int row = int((idx / d_eff) % rows);
        const int64_t n = idx / (int64_t(d_eff) * rows);
        const T* Xn = X + n * int64_t(L) * P.d_in;
        T v = T(0);
        if (mode == MODE_PT_DIFF) {
            v = scaled_point<T>(Xn, L, row, fe, P);
            if (norm_col >= 0) {
                v *= pre;
                if (fe == 0) {
                    T ss = v * v;
                    for (int g = 1; g < d_eff; ++g) {
                        const T u = pre * scaled_point<T>(Xn, L, row, g, P);
                        ss = fma(u, u, ss);
                    }
                    out[n * rec_elems + int64_t(row) * RS + norm_col] = T(-0.5) * ss;
                }
            }
        } else if (row >= 1) {   // leading zero row
            if (mode == MODE_INC && difference) v = scaled_point<T>(Xn, L, row, fe, P) - scaled_point<T>(Xn, L, row - 1, fe, P);
            else v = scaled_point<T>(Xn, L, row - 1, fe, P);
        }
        out[n * rec_elems + int64_t(row) * RS + fe] = v;
    }
}

// Time-major scaled observations for the tensor-vs-sequence kernel: XT[(t * d_eff + fe) * Npad + n].
template <typename T>
__global__ void prep_seq_timemajor_kernel(const T* __restrict__ X, int64_t N, int64_t Npad, int L, ScaleParams P,
                                          T* __restrict__ out) {
    const int d_eff = P.d_eff();
    const int64_t total = int64_t(L) * d_eff * Npad;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t n = idx % Npad;
        const int fe = int((idx / Npad) % d_eff);
        const int t = int(idx / (Npad * d_eff));
        out[idx] = n < N ? scaled_point<T>(X + n * int64_t(L) * P.d_in, L, t, fe, P) : T(0);
    }
}

// Scaled inducing tensors (kernels.py:367-398).  In: Z (lt, T, E, d_eff) with E = 2 for increments else 1.
// Out: ZT[((t * d_eff + fe) * lt + k) * E + e]  and  ZS[(t * lt + k) * E + e] = |z|^2.
template <typename T>
__global__ void prep_tensors_kernel(const T* __restrict__ Z, int lt, int64_t Tn, int E, ScaleParams P,
                                    T* __restrict__ ZT, T* __restrict__ ZS) {
    const int d_eff = P.d_eff();
    const int64_t total = Tn * lt * E;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int e = int(idx % E);
        const int k = int((idx / E) % lt);
        const int64_t t = idx / (int64_t(E) * lt);
        T ss = T(0);
        for (int fe = 0; fe < d_eff; ++fe) {
            const int lag = fe / P.d_in, f = fe - lag * P.d_in;
            T v = Z[((int64_t(k) * Tn + t) * E + e) * d_eff + fe];
            if (P.has_ls) {                                   // kernels.py:374-379 / :391-395: no lag weights without lengthscales
                v = v / T(P.lsv(f));
                if (P.num_lags > 0) v = v * T(P.gamma[lag]);
            }
            ZT[((t * d_eff + fe) * lt + k) * E + e] = v;
            ss = fma(v, v, ss);
        }
        ZS[(t * lt + k) * E + e] = ss;
    }
}

// fac[n][m] = w[m] / sqrt(dlev[n][m] + jitter)   (normalise)   or   w[m]   (dlev == nullptr);
// squared == 1: w[m] / (dlev[n][m] + jitter)  -- the double division of kernels.py:713 + :750
template <typename T>
__global__ void factors_kernel(const T* __restrict__ dlev, int64_t N, int M1, const double* __restrict__ w, double jitter,
                               int squared, T* __restrict__ fac) {
    const int64_t total = N * M1;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int m = int(idx % M1);
        const T wm = w ? T(w[m]) : T(1);
        // (K_m(x, x) >= 0; a float32 higher-order recursion can return a slightly negative one for a sequence whose level values cancel
        // from 1e8 to 10 -- the fuzz sweep's case 153, round 4 -- and sqrt of that is a NaN column where the float64 result is finite: clamped)
        if (!dlev) fac[idx] = wm;
        else if (squared) { const T s = sqrt(fmax(dlev[idx], T(0)) + T(jitter)); fac[idx] = wm / s / s; }
        else fac[idx] = wm / sqrt(fmax(dlev[idx], T(0)) + T(jitter));
    }
}

// out[idx] = sum_m in[m * stride + idx] * w[m]   or   out[m * stride + idx] = in[...] * w[m]
template <typename T>
__global__ void weight_levels_kernel(const T* __restrict__ in, int64_t stride, int M1, const double* __restrict__ w,
                                     int sum_levels, T* __restrict__ out) {
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < stride; idx += int64_t(gridDim.x) * blockDim.x) {
        T acc = T(0);
        for (int m = 0; m < M1; ++m) {
            const T v = in[m * stride + idx] * T(w[m]);
            if (sum_levels) acc += v; else out[m * stride + idx] = v;
        }
        if (sum_levels) out[idx] = acc;
    }
}

// Full symmetric matrix from stacked owned-row blocks (see gpsig_kernel_K_symm_rows): the ownership rule is the
// emission predicate of the seq-gram kernel (PRED_CIRCULANT in seq_args.hpp) with i = column, j = row.
template <typename T>
__global__ void symmetrize_owned_rows_kernel(const T* __restrict__ half, int64_t N, T* __restrict__ out) {
    const int64_t total = N * N, H = N / 2;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t r = idx / N, c = idx - r * N;
        int64_t dlt = r - c;
        if (dlt < 0) dlt += N;
        const bool owned = dlt < H || (dlt == H && ((N & 1) || c < r));
        out[idx] = owned ? half[idx] : half[c * N + r];
    }
}

// The same from COMPACT row blocks (gpsig_kernel_K_symm_rows_compact): half is (N, N/2+1), row r holding its owned columns
// r-N/2 .. r (mod N) side by side, half[r][N/2 - (r-c) mod N].  One 64 x 64 tile of `out` per block iteration: the entries the
// rows of the tile own are copied straight (coalesced along c); the others belong to the tile's columns and are contiguous in
// half along r, so they are read with r varying fastest and turned through LDS.
template <typename T>
__global__ __launch_bounds__(256) void symmetrize_compact_rows_kernel(const T* __restrict__ half, int64_t N, T* __restrict__ out) {
    __shared__ T tile[64][65];
    const int64_t H = N / 2, W = H + 1, tpr = (N + 63) / 64, ntiles = tpr * tpr;
    const int tx = threadIdx.x, ty = threadIdx.y;
    auto owns = [&](int64_t r, int64_t c, int64_t* dlt) {      // does row r own column c
        int64_t d = r - c;
        if (d < 0) d += N;
        *dlt = d;
        return d < H || (d == H && ((N & 1) || c < r));
    };
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t r0 = (t / tpr) * 64, c0 = (t % tpr) * 64;
        for (int k = ty; k < 64; k += 4) {                       // tile[c - c0][r - r0] = half[c][.] where column c owns (r, c)
            const int64_t c = c0 + k, r = r0 + tx;
            int64_t d;
            if (r < N && c < N && r != c && owns(c, r, &d)) tile[k][tx] = half[c * W + H - d];
        }
        __syncthreads();
        for (int k = ty; k < 64; k += 4) {
            const int64_t r = r0 + k, c = c0 + tx;
            int64_t d;
            if (r < N && c < N) out[r * N + c] = owns(r, c, &d) ? half[r * W + H - d] : tile[tx][k];
        }
        __syncthreads();
    }
}

template <typename T>
__global__ void fill_kernel(T* __restrict__ p, int64_t n, T v) {
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < n; idx += int64_t(gridDim.x) * blockDim.x) p[idx] = v;
}

// ---------------------------------------------------------------------------------------------
// Inducing tensors vs sequences: SignatureKernel._K_tens_vs_seq + signature_kern_tens_vs_seq_first_order
// (gpsig/kernels.py:313-340, gpsig/signature_algs.py:101-127) + the epilogue of K_tens_vs_seq
// (kernels.py:572-588).  One lane per sequence, TT tensors per wave; the time axis is swept once with
// the lt = M(M+1)/2 running sums of the level chains in registers:
//     level i uses components k0 .. k0+i-1 (k0 = i(i-1)/2):
//     u[k0+j] += dM[k0+j](tau) * u[k0+j-1]   (j = i-1 .. 1, using u from before this time step)
//     u[k0]   += dM[k0](tau)                  ;  K_i = u[k0+i-1] after the last step
// which is the exclusive-cumsum chain of signature_algs.py:120-125.  No cross-lane traffic; the
// sequences are read time-major (coalesced over lanes), the tensor components are wave-uniform.
struct TvsArgs {
    const void* XT;     // (L, d_eff, Npad) scaled observations
    const void* ZT;     // (T, d_eff, lt, E) scaled tensor components
    const void* ZS;     // (T, lt, E) squared norms
    int64_t N, Npad, Tn;
    int32_t L, d_eff, kind, difference, order;
    double p0, p1;
    const void* fx;     // (N, M+1) per-sequence factors (1/sqrt(diag+jitter)) or NULL
    const double* w;    // (M+1) level weights sigma*variances, or NULL (raw levels)
    void* out;          // (T, N) or (M+1, T, N)
    int32_t sum_levels;
    const double* spec; // BASE_SPECTRAL table (seq_core.hpp: spectral_eval); p0 = Q, p1 = family
};

template <typename T, int M, int TT, bool INCR>
__global__ __launch_bounds__(64) void tens_vs_seq_kernel(const TvsArgs A) {
    constexpr int LT = M * (M + 1) / 2;
    constexpr int E = INCR ? 2 : 1;
    const int lane = threadIdx.x;
    const int64_t n = blockIdx.x * int64_t(64) + lane;
    const int64_t t0 = blockIdx.y * int64_t(TT);
    const T* __restrict__ XT = static_cast<const T*>(A.XT);
    const T* __restrict__ ZT = static_cast<const T*>(A.ZT);
    const T* __restrict__ ZS = static_cast<const T*>(A.ZS);
    const int d = A.d_eff;
    const T p0 = T(A.p0), p1 = T(A.p1);

    T u[TT][LT], kprev[TT][LT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int k = 0; k < LT; ++k) { u[tt][k] = T(0); kprev[tt][k] = T(0); }

    for (int tau = 0; tau < A.L; ++tau) {
        T ip[TT][LT][E];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int k = 0; k < LT; ++k)
#pragma unroll
                for (int e = 0; e < E; ++e) ip[tt][k][e] = T(0);
        T xs = T(0);
        const bool spectral = A.kind == BASE_SPECTRAL;
        if (spectral) {
            // not a function of inner products: every (component, point) takes its own pass over the features
            const T* __restrict__ xcol = XT + int64_t(tau) * d * A.Npad + n;
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const int64_t t = (t0 + tt < A.Tn) ? t0 + tt : A.Tn - 1;
                for (int ke = 0; ke < LT * E; ++ke) {
                    const T* __restrict__ z = ZT + t * d * (LT * E) + ke;
                    const T v = spectral_eval<T>(A.spec, int(A.p0), int(A.p1), d, [&](int f) { return xcol[int64_t(f) * A.Npad]; },
                                                 [&](int f) { return z[int64_t(f) * (LT * E)]; });
#pragma unroll
                    for (int k = 0; k < LT; ++k)
#pragma unroll
                        for (int e = 0; e < E; ++e)
                            if (k * E + e == ke) ip[tt][k][e] = v;
                }
            }
        } else
        for (int f = 0; f < d; ++f) {
            const T x = XT[(int64_t(tau) * d + f) * A.Npad + n];
            xs = fma(x, x, xs);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const int64_t t = (t0 + tt < A.Tn) ? t0 + tt : A.Tn - 1;
                const T* __restrict__ z = ZT + (t * d + f) * (LT * E);     // wave-uniform
#pragma unroll
                for (int k = 0; k < LT; ++k)
#pragma unroll
                    for (int e = 0; e < E; ++e) ip[tt][k][e] = fma(z[k * E + e], x, ip[tt][k][e]);
            }
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int64_t t = (t0 + tt < A.Tn) ? t0 + tt : A.Tn - 1;
            const T* __restrict__ zs = ZS + t * (LT * E);
            T dm[LT];
            {
                T kvv[LT * E], zz[LT * E];
#pragma unroll
                for (int k = 0; k < LT; ++k)
#pragma unroll
                    for (int e = 0; e < E; ++e) { kvv[k * E + e] = ip[tt][k][e]; zz[k * E + e] = zs[k * E + e]; }
                if (!spectral) base_eval_n<T, LT * E>(A.kind, kvv, zz, xs, p0, p1);
#pragma unroll
                for (int k = 0; k < LT; ++k) {
                    const T kv = INCR ? kvv[k * E + E - 1] - kvv[k * E] : kvv[k * E];   // kernels.py:329-330
                    if (A.difference) { dm[k] = kv - kprev[tt][k]; kprev[tt][k] = kv; }   // signature_algs.py:114
                    else dm[k] = kv;
                }
            }
            if (!(A.difference && tau == 0)) {
                if (A.order <= 1) {
#pragma unroll
                    for (int i = M; i >= 1; --i) {
                        const int k0 = i * (i - 1) / 2;
#pragma unroll
                        for (int j = i - 1; j >= 1; --j) u[tt][k0 + j] = fma(dm[k0 + j], u[tt][k0 + j - 1], u[tt][k0 + j]);
                        u[tt][k0] += dm[k0];
                    }
                } else {
                    // higher-order chains (signature_algs.py:147-158): position j of level i keeps min(j+1, order) terms
                    // R[l] of this time step; R'[0] = dM * excumsum(sum R), R'[l] = dM * R[l-1] / (l+1).
#pragma unroll
                    for (int i = 1; i <= M; ++i) {
                        const int k0 = i * (i - 1) / 2;
                        T Rp[M], Rc[M];
                        Rp[0] = dm[k0];
                        T totp = dm[k0];
#pragma unroll
                        for (int j = 1; j < i; ++j) {
                            const int dj = (j + 1 < A.order) ? j + 1 : A.order;
                            Rc[0] = dm[k0 + j] * u[tt][k0 + j - 1];                       // running sum up to the previous step
                            T totc = Rc[0];
#pragma unroll
                            for (int l = 1; l <= j; ++l)
                                if (l < dj) { Rc[l] = (dm[k0 + j] * (T(1) / T(l + 1))) * Rp[l - 1]; totc += Rc[l]; }
                            u[tt][k0 + j - 1] += totp;
#pragma unroll
                            for (int l = 0; l <= j; ++l) Rp[l] = Rc[l];
                            totp = totc;
                        }
                        u[tt][k0 + i - 1] += totp;
                    }
                }
            }
        }
    }

    if (n >= A.N) return;
    const T* fx = A.fx ? static_cast<const T*>(A.fx) + n * (M + 1) : nullptr;
    T* out = static_cast<T*>(A.out);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int64_t t = t0 + tt;
        if (t >= A.Tn) break;
        T acc = T(0);
#pragma unroll
        for (int i = 0; i <= M; ++i) {
            T v = i == 0 ? T(1) : u[tt][i * (i - 1) / 2 + i - 1];
            if (fx) v *= fx[i];
            if (A.w) v *= T(A.w[i]);
            if (A.sum_levels) acc += v;
            else out[(int64_t(i) * A.Tn + t) * A.N + n] = v;
        }
        if (A.sum_levels) out[t * A.N + n] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Tensor-lane variant of the tensor-vs-sequence kernel: one lane per inducing TENSOR, one wavefront per
// (block of 64 tensors, sequence).  The sequence's observation x(tau) is then wave-uniform (scalar loads of
// the caller's own (N, L, d) layout after scaling, no transposed copy) and the tensor components of the level
// being swept sit in registers, so the inner product is D fp64 FMAs with a scalar operand and no LDS or
// per-lane memory traffic inside the time loop.  Levels are swept one after the other (a level's chain only
// involves its own components, signature_algs.py:118-125).  Used when there are at least 32 tensors.
struct TvsLaneTArgs {
    const void* XS;     // (N, L, d_eff) scaled observations, sequence-major
    const void* ZL;     // (lt, E, d_eff, Tpad) scaled tensor components, tensor index fastest
    const void* ZN;     // (lt, E, Tpad) squared norms
    int64_t N, Tn, Tpad;
    int32_t L, d_eff, kind, difference, order, M;
    double p0, p1;
    const void* fx;     // (N, M+1) per-sequence factors or NULL
    const double* w;    // (M+1) level weights or NULL
    void* out;          // (T, N) or (M+1, T, N)
    int32_t sum_levels;
};

// One kernel instance sweeps the levels LO..HI together (their components all in registers: the host splits the
// levels into groups that fit, tvl_plan in tens_inst.hip); the first group also writes level 0, later groups add
// to the level sum written by the earlier ones (same stream, so ordered).
template <typename T, int LO, int HI, int D, bool INCR>
__global__ __launch_bounds__(64) void tens_vs_seq_lanet_kernel(const TvsLaneTArgs A) {
    constexpr int E = INCR ? 2 : 1;
    constexpr int K0 = LO * (LO - 1) / 2;                         // first component of level LO
    constexpr int NC = HI * (HI + 1) / 2 - K0;                    // components of levels LO..HI
    extern __shared__ __attribute__((aligned(16))) unsigned char tvl_smem[];
    T* const xbuf = reinterpret_cast<T*>(tvl_smem);               // the whole scaled sequence: L * d values
    const int lane = threadIdx.x;
    const int64_t t = blockIdx.x * int64_t(64) + lane;            // < Tpad by construction
    const int64_t n = blockIdx.y;
    const int d = A.d_eff, M = A.M;
    {   // one coalesced pass over the sequence; every later read is a same-address (broadcast) LDS read
        const T* __restrict__ XG = static_cast<const T*>(A.XS) + n * int64_t(A.L) * d;
        for (int idx = lane; idx < A.L * d; idx += 64) xbuf[idx] = XG[idx];
    }
    const T* __restrict__ ZL = static_cast<const T*>(A.ZL);
    const T* __restrict__ ZN = static_cast<const T*>(A.ZN);
    const T p0 = T(A.p0), p1 = T(A.p1);
    const T* fx = A.fx ? static_cast<const T*>(A.fx) + n * (M + 1) : nullptr;
    T* out = static_cast<T*>(A.out);

    T z[NC][E][D], zz[NC * E];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            zz[c * E + e] = ZN[(int64_t(K0 + c) * E + e) * A.Tpad + t];
#pragma unroll
            for (int f = 0; f < D; ++f) z[c][e][f] = (f < d) ? ZL[((int64_t(K0 + c) * E + e) * d + f) * A.Tpad + t] : T(0);
        }
    T u[NC], kprev[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { u[c] = T(0); kprev[c] = T(0); }

    for (int tau = 0; tau < A.L; ++tau) {
        T x[D];
        T xs = T(0);
#pragma unroll
        for (int f = 0; f < D; ++f) { x[f] = (f < d) ? xbuf[tau * d + f] : T(0); xs = fma(x[f], x[f], xs); }
        T kvv[NC * E];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                T ip = z[c][e][0] * x[0];
#pragma unroll
                for (int f = 1; f < D; ++f) ip = fma(z[c][e][f], x[f], ip);
                kvv[c * E + e] = ip;
            }
        base_eval_n<T, NC * E>(A.kind, kvv, zz, xs, p0, p1);
        T dm[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const T k1 = INCR ? kvv[c * E + E - 1] - kvv[c * E] : kvv[c * E];               // kernels.py:329-330
            if (A.difference) { dm[c] = k1 - kprev[c]; kprev[c] = k1; } else dm[c] = k1;     // signature_algs.py:114
        }
        if (A.difference && tau == 0) continue;
#pragma unroll
        for (int i = LO; i <= HI; ++i) {
            constexpr int dummy0 = 0; (void)dummy0;
            const int c0 = i * (i - 1) / 2 - K0;
            if (A.order <= 1) {
#pragma unroll
                for (int j = i - 1; j >= 1; --j) u[c0 + j] = fma(dm[c0 + j], u[c0 + j - 1], u[c0 + j]);   // signature_algs.py:122-124
                u[c0] += dm[c0];
            } else {                                                                         // signature_algs.py:147-158
                T Rp[HI], Rc[HI];
                Rp[0] = dm[c0];
                T totp = dm[c0];
#pragma unroll
                for (int j = 1; j < i; ++j) {
                    const int dj = (j + 1 < A.order) ? j + 1 : A.order;
                    Rc[0] = dm[c0 + j] * u[c0 + j - 1];
                    T totc = Rc[0];
#pragma unroll
                    for (int l = 1; l <= j; ++l)
                        if (l < dj) { Rc[l] = (dm[c0 + j] * (T(1) / T(l + 1))) * Rp[l - 1]; totc += Rc[l]; }
                    u[c0 + j - 1] += totp;
#pragma unroll
                    for (int l = 0; l <= j; ++l) Rp[l] = Rc[l];
                    totp = totc;
                }
                u[c0 + i - 1] += totp;
            }
        }
    }

    if (t >= A.Tn) return;
    T acc = T(0);
    if (LO == 1) {                                                 // level 0 (signature_algs.py:116)
        T v0 = T(1);
        if (fx) v0 *= fx[0];
        if (A.w) v0 *= T(A.w[0]);
        if (A.sum_levels) acc = v0; else out[(int64_t(0) * A.Tn + t) * A.N + n] = v0;
    } else if (A.sum_levels) {
        acc = out[t * A.N + n];
    }
#pragma unroll
    for (int i = LO; i <= HI; ++i) {
        T v = u[i * (i - 1) / 2 - K0 + i - 1];
        if (fx) v *= fx[i];
        if (A.w) v *= T(A.w[i]);
        if (A.sum_levels) acc += v; else out[(int64_t(i) * A.Tn + t) * A.N + n] = v;
    }
    if (A.sum_levels) out[t * A.N + n] = acc;
}

// Scaled inducing tensors in the tensor-lane layout.  In: Z (lt, T, E, d_eff).
// Out: ZL[((k * E + e) * d_eff + fe) * Tpad + t], ZN[(k * E + e) * Tpad + t] = |z|^2  (zero for t >= T).
template <typename T>
__global__ void prep_tensors_lanet_kernel(const T* __restrict__ Z, int lt, int64_t Tn, int64_t Tpad, int E, ScaleParams P,
                                          T* __restrict__ ZL, T* __restrict__ ZN) {
    const int d_eff = P.d_eff();
    const int64_t total = Tpad * lt * E;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t t = idx % Tpad;
        const int e = int((idx / Tpad) % E);
        const int k = int(idx / (Tpad * E));
        T ss = T(0);
        for (int fe = 0; fe < d_eff; ++fe) {
            const int lag = fe / P.d_in, f = fe - lag * P.d_in;
            T v = T(0);
            if (t < Tn) {
                v = Z[((int64_t(k) * Tn + t) * E + e) * d_eff + fe];
                if (P.has_ls) {
                    v = v / T(P.lsv(f));
                    if (P.num_lags > 0) v = v * T(P.gamma[lag]);
                }
            }
            ZL[((int64_t(k) * E + e) * d_eff + fe) * Tpad + t] = v;
            ss = fma(v, v, ss);
        }
        ZN[(int64_t(k) * E + e) * Tpad + t] = ss;
    }
}

// Scaled observations in the caller's own layout: XS[(n * L + t) * d_eff + fe].
template <typename T>
__global__ void prep_seq_scaled_kernel(const T* __restrict__ X, int64_t N, int L, ScaleParams P, T* __restrict__ out) {
    const int d_eff = P.d_eff();
    const int64_t total = N * int64_t(L) * d_eff;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int fe = int(idx % d_eff);
        const int t = int((idx / d_eff) % L);
        const int64_t n = idx / (int64_t(d_eff) * L);
        out[idx] = scaled_point<T>(X + n * int64_t(L) * P.d_in, L, t, fe, P);
    }
}

// ---------------------------------------------------------------------------------------------
// Inducing tensors vs inducing tensors: SignatureKernel._K_tens + tensor_kern (kernels.py:263-283,
// signature_algs.py:76-99) + sigma*variances weighting (kernels.py:531-536).  One thread per (t, t').
struct TensGramArgs {
    const void* ZT;   // (T, d_eff, lt, E)
    const void* ZS;   // (T, lt, E)
    int64_t Tn;
    int32_t M, d_eff, E, kind;
    double p0, p1;
    const double* w;  // (M+1) or NULL
    void* out;        // (T, T) or (M+1, T, T)
    int32_t sum_levels;
    const double* spec;   // BASE_SPECTRAL table
};

template <typename T>
__global__ void tens_gram_kernel(const TensGramArgs A) {
    const int64_t total = A.Tn * A.Tn;
    const T* __restrict__ ZT = static_cast<const T*>(A.ZT);
    const T* __restrict__ ZS = static_cast<const T*>(A.ZS);
    T* out = static_cast<T*>(A.out);
    const int lt = A.M * (A.M + 1) / 2, E = A.E, d = A.d_eff;
    const T p0 = T(A.p0), p1 = T(A.p1);
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t t1 = idx / A.Tn, t2 = idx % A.Tn;
        T acc = T(0);
        int k = 0;
        for (int i = 0; i <= A.M; ++i) {
            T R = T(1);
            for (int j = 0; j < i; ++j, ++k) {
                T mk;
                if (E == 1) {
                    if (A.kind == BASE_SPECTRAL) {
                        mk = spectral_eval<T>(A.spec, int(A.p0), int(A.p1), d, [&](int f) { return ZT[(t1 * d + f) * lt + k]; },
                                              [&](int f) { return ZT[(t2 * d + f) * lt + k]; });
                    } else {
                        T ip = T(0);
                        for (int f = 0; f < d; ++f) ip = fma(ZT[(t1 * d + f) * lt + k], ZT[(t2 * d + f) * lt + k], ip);
                        mk = base_eval<T>(A.kind, ip, ZS[t1 * lt + k], ZS[t2 * lt + k], p0, p1);
                    }
                } else {   // kernels.py:275-277
                    T kv[2][2];
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) {
                            if (A.kind == BASE_SPECTRAL) {
                                kv[a][b] = spectral_eval<T>(A.spec, int(A.p0), int(A.p1), d, [&](int f) { return ZT[((t1 * d + f) * lt + k) * 2 + a]; },
                                                            [&](int f) { return ZT[((t2 * d + f) * lt + k) * 2 + b]; });
                                continue;
                            }
                            T ip = T(0);
                            for (int f = 0; f < d; ++f)
                                ip = fma(ZT[((t1 * d + f) * lt + k) * 2 + a], ZT[((t2 * d + f) * lt + k) * 2 + b], ip);
                            kv[a][b] = base_eval<T>(A.kind, ip, ZS[(t1 * lt + k) * 2 + a], ZS[(t2 * lt + k) * 2 + b], p0, p1);
                        }
                    mk = kv[1][1] + kv[0][0] - kv[1][0] - kv[0][1];
                }
                R = (j == 0) ? mk : mk * R;                       // signature_algs.py:92-96
            }
            T v = R;
            if (A.w) v *= T(A.w[i]);
            if (A.sum_levels) acc += v; else out[int64_t(i) * total + idx] = v;
        }
        if (A.sum_levels) out[idx] = acc;
    }
}

// The same in tiles of 16 x 16 entries per workgroup: the 16 + 16 tensors a tile touches (contiguous records of
// d_eff * lt * E components + lt * E squared norms each) are copied to LDS once, coalesced, instead of every thread
// gathering its 2 * lt * E * d_eff components from HBM at a stride of one record per lane (Kzz at T = 512, M = 4, d = 6:
// 101 -> about 10 us).  Row stride of the LDS copies odd: the 16 different t2 of a wavefront hit 16 different banks, the 4
// different t1 are broadcasts.  Not for the spectral kernel (it takes the points through its own accessors).
constexpr int TENS_TILE = 16;
inline int tens_gram_tile_stride(int d_eff, int lt, int E) { return (d_eff * lt * E + lt * E) | 1; }
template <typename T>
__global__ __launch_bounds__(TENS_TILE * TENS_TILE) void tens_gram_tile_kernel(const TensGramArgs A, int RSZ) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tens_tile_lds[];
    T* const za = reinterpret_cast<T*>(tens_tile_lds);          // [16][RSZ]: components then squared norms of tensors ta ..
    T* const zb = za + TENS_TILE * RSZ;                          //            the same for tb ..
    const T* __restrict__ ZT = static_cast<const T*>(A.ZT);
    const T* __restrict__ ZS = static_cast<const T*>(A.ZS);
    T* out = static_cast<T*>(A.out);
    const int lt = A.M * (A.M + 1) / 2, E = A.E, d = A.d_eff;
    const int nc = d * lt * E, ns = lt * E;
    const int64_t ta = int64_t(blockIdx.y) * TENS_TILE, tb = int64_t(blockIdx.x) * TENS_TILE;
    for (int q = threadIdx.x; q < TENS_TILE * (nc + ns); q += TENS_TILE * TENS_TILE) {
        const int r = q / (nc + ns), o = q - r * (nc + ns);
        const int64_t t1 = ta + r < A.Tn ? ta + r : A.Tn - 1, t2 = tb + r < A.Tn ? tb + r : A.Tn - 1;
        za[r * RSZ + o] = o < nc ? ZT[t1 * nc + o] : ZS[t1 * ns + (o - nc)];
        zb[r * RSZ + o] = o < nc ? ZT[t2 * nc + o] : ZS[t2 * ns + (o - nc)];
    }
    __syncthreads();
    const int ty = threadIdx.x / TENS_TILE, tx = threadIdx.x % TENS_TILE;
    const int64_t t1 = ta + ty, t2 = tb + tx;
    if (t1 >= A.Tn || t2 >= A.Tn) return;
    const T* a = za + ty * RSZ;
    const T* b = zb + tx * RSZ;
    const int64_t total = A.Tn * A.Tn, idx = t1 * A.Tn + t2;
    const T p0 = T(A.p0), p1 = T(A.p1);
    T acc = T(0);
    int k = 0;
    for (int i = 0; i <= A.M; ++i) {
        T R = T(1);
        for (int j = 0; j < i; ++j, ++k) {
            T mk;
            if (E == 1) {
                T ip = T(0);
                for (int f = 0; f < d; ++f) ip = fma(a[f * lt + k], b[f * lt + k], ip);
                mk = base_eval<T>(A.kind, ip, a[nc + k], b[nc + k], p0, p1);
            } else {   // kernels.py:275-277
                T kv[2][2];
                for (int u = 0; u < 2; ++u)
                    for (int v = 0; v < 2; ++v) {
                        T ip = T(0);
                        for (int f = 0; f < d; ++f) ip = fma(a[(f * lt + k) * 2 + u], b[(f * lt + k) * 2 + v], ip);
                        kv[u][v] = base_eval<T>(A.kind, ip, a[nc + k * 2 + u], b[nc + k * 2 + v], p0, p1);
                    }
                mk = kv[1][1] + kv[0][0] - kv[1][0] - kv[0][1];
            }
            R = (j == 0) ? mk : mk * R;                           // signature_algs.py:92-96
        }
        T v = R;
        if (A.w) v *= T(A.w[i]);
        if (A.sum_levels) acc += v; else out[int64_t(i) * total + idx] = v;
    }
    if (A.sum_levels) out[idx] = acc;
}

// ---- any-shape fallback of the sequence-vs-sequence recursion (first-order algorithm, signature_algs.py:8-35) -------------
// One pair per thread, 64 consecutive x sequences against one y sequence per wavefront (or the pairs (i, i) for the
// diagonal).  The previous lattice row of Q_1..Q_{M-1} sits in an HBM scratch array laid out [level][column][pair], so a
// wavefront's accesses are contiguous; nothing has to fit registers or LDS, hence no limit on either length or on the
// number of features.  Orders of magnitude slower per pair than seq_gram_kernel: used only for shapes that kernel is not
// built for (register-side sequences beyond its column capacity, more than 32 state-space dimensions after lags).
struct GenericSeqArgs {
    const double* XT; const double* YT;    // time-major scaled points [(t * d + f) * stride + n]
    int64_t xstride, ystride;
    int64_t N1, N2;
    int32_t L1, L2, d, M, kind, mode, diag;
    double p0, p1;
    const double* spec;
    int64_t j0;                            // this launch covers y sequences j0 .. j0 + gridDim.y - 1
    double* scratch;                       // (M-1) * R2 * pairs doubles
    int64_t pairs;
    double* out;                           // levels: out[m * sm + i * si + j * sj]
    int64_t sm, si, sj;
};

__device__ __forceinline__ double generic_kappa(const GenericSeqArgs& A, int64_t i, int ta, int64_t j, int tb) {
    const double* x = A.XT + int64_t(ta) * A.d * A.xstride + i;
    const double* y = A.YT + int64_t(tb) * A.d * A.ystride + j;
    if (A.kind == BASE_SPECTRAL)
        return spectral_eval<double>(A.spec, int(A.p0), int(A.p1), A.d, [&](int f) { return x[int64_t(f) * A.xstride]; },
                                     [&](int f) { return y[int64_t(f) * A.ystride]; });
    double in = 0.0, xs = 0.0, ys = 0.0;
    for (int f = 0; f < A.d; ++f) {
        const double xv = x[int64_t(f) * A.xstride], yv = y[int64_t(f) * A.ystride];
        in = fma(xv, yv, in); xs = fma(xv, xv, xs); ys = fma(yv, yv, ys);
    }
    return base_eval<double>(A.kind, in, xs, ys, A.p0, A.p1);
}

// grid (ceil(N1 / 64), number of y sequences of this launch or 1 for the diagonal); block 64
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void __launch_bounds__(64) seq_levels_generic_kernel(const GenericSeqArgs A) {
    constexpr int MM = 8;
    const int64_t i = int64_t(blockIdx.x) * 64 + threadIdx.x;
    const bool valid = i < A.N1;
    const int64_t ii = valid ? i : 0;
    const int64_t j = A.diag ? ii : A.j0 + blockIdx.y;
    const int64_t pidx = (int64_t(blockIdx.y) * gridDim.x + blockIdx.x) * 64 + threadIdx.x;
    const int dr = A.mode == MODE_PT_NODIFF ? 0 : 1;
    const int R1 = A.L1 - dr, R2 = A.L2 - dr, M = A.M;
    auto qrow = [&](int m, int b) -> double& { return A.scratch[(int64_t(m - 1) * R2 + b) * A.pairs + pidx]; };
    double ktop = 0.0, qlast[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) qlast[m] = 0.0;
    for (int a = 0; a < R1; ++a) {
        double s[MM + 1], qd[MM];
#pragma unroll
        for (int m = 0; m <= MM; ++m) s[m] = 0.0;
#pragma unroll
        for (int m = 0; m < MM; ++m) qd[m] = 0.0;
        double klo = 0.0, khi = 0.0;            // kappa(x_a, y_b), kappa(x_{a+1}, y_b)
        if (A.mode == MODE_PT_DIFF) { klo = generic_kappa(A, ii, a, j, 0); khi = generic_kappa(A, ii, a + 1, j, 0); }
        for (int b = 0; b < R2; ++b) {
            double dm;
            if (A.mode == MODE_INC) {
                dm = 0.0;
                for (int f = 0; f < A.d; ++f) {
                    const double* x = A.XT + (int64_t(a) * A.d + f) * A.xstride + ii;
                    const double* y = A.YT + (int64_t(b) * A.d + f) * A.ystride + j;
                    dm = fma(x[int64_t(A.d) * A.xstride] - x[0], y[int64_t(A.d) * A.ystride] - y[0], dm);
                }
            } else if (A.mode == MODE_PT_DIFF) {
                const double nlo = generic_kappa(A, ii, a, j, b + 1), nhi = generic_kappa(A, ii, a + 1, j, b + 1);
                dm = (nhi - khi) - (nlo - klo);                  // signature_algs.py:26
                klo = nlo; khi = nhi;
            } else {
                dm = generic_kappa(A, ii, a, j, b);
            }
            double qup[MM];
#pragma unroll
            for (int m = 1; m < MM; ++m) qup[m] = (m < M && a > 0) ? qrow(m, b) : 0.0;
            s[1] += dm;
#pragma unroll
            for (int m = 2; m <= MM; ++m)
                if (m <= M) s[m] = fma(dm, qd[m - 1], s[m]);
#pragma unroll
            for (int m = 1; m < MM; ++m)
                if (m < M) {
                    const double q = qup[m] + s[m];
                    qrow(m, b) = q;
                    qd[m] = qup[m];
                    qlast[m] = q;
                }
        }
#pragma unroll
        for (int m = 1; m <= MM; ++m)
            if (m == M) ktop += s[m];
    }
    if (!valid) return;
    double* o = A.out + i * A.si + (A.diag ? 0 : j * A.sj);
    o[0] = 1.0;
#pragma unroll
    for (int m = 1; m < MM; ++m)
        if (m < M) o[m * A.sm] = (R1 > 0 && R2 > 0) ? qlast[m] : 0.0;
    o[int64_t(M) * A.sm] = ktop;
}
#else
__global__ void __launch_bounds__(64) seq_levels_generic_kernel(const GenericSeqArgs A);
#endif

// Higher-order algorithm (signature_algs.py:37-74) in the same any-shape, one-pair-per-thread form.  Per lattice cell and
// level m the reference's d x d grid (d = min(m, order)) is
//     R_m[0][0]     = dM * (exclusive 2-D prefix of the grid total of level m-1)                       (:64)
//     R_m[0][j-1]   = dM / j * (exclusive prefix over rows a of the column sum  sum_r R_{m-1}[r][j-2])   (:66)
//     R_m[j-1][0]   = dM / j * (exclusive prefix over columns b of the row sum  sum_s R_{m-1}[j-2][s])   (:67)
//     R_m[j-1][k-1] = dM / (j k) * R_{m-1}[j-2][k-2]   at the same cell                                    (:69)
// and K_m = sum over cells and grid entries (:71).  Scratch per pair and lattice column: for each level m < M the inclusive
// 2-D prefix Q_m of the previous row and the OM column prefixes; the row prefixes run in (private) registers along a row.
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void __launch_bounds__(64) seq_levels_generic_ho_kernel(const GenericSeqArgs A, int order) {
    constexpr int MM = 8, OM = 8;
    const int64_t i = int64_t(blockIdx.x) * 64 + threadIdx.x;
    const bool valid = i < A.N1;
    const int64_t ii = valid ? i : 0;
    const int64_t j = A.diag ? ii : A.j0 + blockIdx.y;
    const int64_t pidx = (int64_t(blockIdx.y) * gridDim.x + blockIdx.x) * 64 + threadIdx.x;
    const int dr = A.mode == MODE_PT_NODIFF ? 0 : 1;
    const int R1 = A.L1 - dr, R2 = A.L2 - dr, M = A.M;
    // scratch slot (level m in 1..M-1, q in 0..OM, column b): q == 0 -> Q_m, q == 1 + s -> column prefix of grid column s
    auto slot = [&](int m, int q, int b) -> double& { return A.scratch[((int64_t(m - 1) * (OM + 1) + q) * R2 + b) * A.pairs + pidx]; };
    double K[MM + 1];
    for (int m = 0; m <= MM; ++m) K[m] = 0.0;
    for (int a = 0; a < R1; ++a) {
        double sq[MM], qd[MM], RE[MM][OM];       // row prefix of the grid total, Q_m[a-1][b-1], row prefixes of the row sums
        for (int m = 0; m < MM; ++m) {
            sq[m] = qd[m] = 0.0;
            for (int r = 0; r < OM; ++r) RE[m][r] = 0.0;
        }
        double klo = 0.0, khi = 0.0;
        if (A.mode == MODE_PT_DIFF) { klo = generic_kappa(A, ii, a, j, 0); khi = generic_kappa(A, ii, a + 1, j, 0); }
        for (int b = 0; b < R2; ++b) {
            double dm;
            if (A.mode == MODE_INC) {
                dm = 0.0;
                for (int f = 0; f < A.d; ++f) {
                    const double* x = A.XT + (int64_t(a) * A.d + f) * A.xstride + ii;
                    const double* y = A.YT + (int64_t(b) * A.d + f) * A.ystride + j;
                    dm = fma(x[int64_t(A.d) * A.xstride] - x[0], y[int64_t(A.d) * A.ystride] - y[0], dm);
                }
            } else if (A.mode == MODE_PT_DIFF) {
                const double nlo = generic_kappa(A, ii, a, j, b + 1), nhi = generic_kappa(A, ii, a + 1, j, b + 1);
                dm = (nhi - khi) - (nlo - klo);
                klo = nlo; khi = nhi;
            } else {
                dm = generic_kappa(A, ii, a, j, b);
            }
            double Rp[OM][OM], Rc[OM][OM];        // grids of level m-1 and m at this cell
            Rp[0][0] = dm;                        // level 1 (:58-60)
            int dp = 1;
            K[1] += dm;
            for (int m = 2; m <= M + 1; ++m) {
                const int lv = m - 1;             // level whose grid is in Rp
                // state of level lv before this cell: exclusive prefixes
                double qup = 0.0, CE[OM];
                for (int s2 = 0; s2 < OM; ++s2) CE[s2] = 0.0;
                if (lv < M) {
                    if (a > 0) {
                        qup = slot(lv, 0, b);
                        for (int s2 = 0; s2 < dp; ++s2) CE[s2] = slot(lv, 1 + s2, b);
                    }
                }
                if (m <= M) {
                    const int dc = m < order ? m : order;
                    Rc[0][0] = dm * qd[lv - 1];                                               // Q_lv[a-1][b-1]
                    for (int jj = 2; jj <= dc; ++jj) {
                        Rc[0][jj - 1] = dm / jj * CE[jj - 2];
                        Rc[jj - 1][0] = dm / jj * RE[lv - 1][jj - 2];
                        for (int kk = 2; kk <= dc; ++kk) Rc[jj - 1][kk - 1] = dm / (double(jj) * kk) * Rp[jj - 2][kk - 2];
                    }
                    double tot = 0.0;
                    for (int r = 0; r < dc; ++r)
                        for (int s2 = 0; s2 < dc; ++s2) tot += Rc[r][s2];
                    K[m] += tot;
                }
                // fold this cell of level lv into its prefixes (level lv + 1 has consumed the exclusive values)
                if (lv < M) {
                    double tot = 0.0;
                    for (int r = 0; r < dp; ++r) {
                        double rs = 0.0;
                        for (int s2 = 0; s2 < dp; ++s2) rs += Rp[r][s2];
                        RE[lv - 1][r] += rs;
                        tot += rs;
                    }
                    for (int s2 = 0; s2 < dp; ++s2) {
                        double cs = 0.0;
                        for (int r = 0; r < dp; ++r) cs += Rp[r][s2];
                        slot(lv, 1 + s2, b) = CE[s2] + cs;
                    }
                    sq[lv - 1] += tot;
                    slot(lv, 0, b) = qup + sq[lv - 1];
                    qd[lv - 1] = qup;
                }
                if (m <= M) {
                    const int dc = m < order ? m : order;
                    for (int r = 0; r < dc; ++r)
                        for (int s2 = 0; s2 < dc; ++s2) Rp[r][s2] = Rc[r][s2];
                    dp = dc;
                }
            }
        }
    }
    if (!valid) return;
    double* o = A.out + i * A.si + (A.diag ? 0 : j * A.sj);
    o[0] = 1.0;
    for (int m = 1; m <= M; ++m) o[m * A.sm] = K[m];
}
#else
__global__ void __launch_bounds__(64) seq_levels_generic_ho_kernel(const GenericSeqArgs A, int order);
#endif

// levels (M1, N1, N2) -> out[i][j] = sum_m (lev + jitter_diag * [i == j]) * ax[i][m] * by[j][m]   (or per level)
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void levels_epilogue_kernel(const double* __restrict__ lev, int64_t N1, int64_t N2, int M1, const double* __restrict__ ax,
                                       const double* __restrict__ by, double jitter_diag, int sum_levels, double* __restrict__ out) {
    const int64_t total = N1 * N2;
    for (int64_t idx = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t i = idx / N2, j = idx % N2;
        double acc = 0.0;
        for (int m = 0; m < M1; ++m) {
            double v = lev[m * total + idx] + (i == j ? jitter_diag : 0.0);
            if (ax) v *= ax[i * M1 + m];
            if (by) v *= by[j * M1 + m];
            if (sum_levels) acc += v; else out[m * total + idx] = v;
        }
        if (sum_levels) out[idx] = acc;
    }
}
#else
__global__ void levels_epilogue_kernel(const double* __restrict__ lev, int64_t N1, int64_t N2, int M1, const double* __restrict__ ax,
                                       const double* __restrict__ by, double jitter_diag, int sum_levels, double* __restrict__ out);
#endif

}  // namespace gpsig
