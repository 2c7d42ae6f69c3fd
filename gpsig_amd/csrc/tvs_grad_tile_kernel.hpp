// tvs_grad_tile_kernel.hpp -- reverse pass of the tensor-vs-sequence chains (Kzx) on the plan of the forward tile kernel
// (tvs_tile_kernel.hpp): what TensorFlow's autodiff returns for SignatureKernel._K_tens_vs_seq +
// signature_kern_tens_vs_seq_first_order (gpsig/kernels.py:313-340, gpsig/signature_algs.py:101-127; trained through at
// gpsig/models.py:40-73), order 1, float64.
//
// Mapping.  A workgroup is ONE wavefront: 64 inducing tensors (lane = tensor; with incremental tensors of a non-linear base
// kernel, 32: lanes 2t and 2t+1 hold the two points of tensor t, kernels.py:328-330), a RUN of consecutive sequences, and one of NR
// ROLES = a subset of the levels (a level's chain involves only its own components, signature_algs.py:118-125; the roles of a
// (tensor block, run) are separate workgroups, blockIdx.z, heaviest first), so that a lane keeps the components of its levels AND
// their gradient accumulators in registers for the whole run.  A sequence is staged into LDS by LDS-DMA (double-buffered) and read
// back as same-address broadcasts.  Per sequence the wave makes TWO sweeps over time for ALL its levels at once:
//   forward   u_{j+1}[tau] = u_{j+1}[tau-1] + m_j[tau] u_j[tau-1]     (the chains of signature_algs.py:120-124, u_0 == 1)
//   backward  tau = last .. first: the chain prefixes are rebuilt by undoing the forward step (u_{j+1}[tau-1] = u_{j+1}[tau] -
//             m_j[tau] u_j[tau-1], lowest chain first), W_j[tau] = dL/du_j[tau] runs alongside (W_i == the upstream gradient of
//             level i), dL/dm_j[tau] = u_j[tau-1] W_{j+1}[tau], and the difference along time (signature_algs.py:114) turns that
//             into dL/dkappa_j(x_tau), which is contracted with the base kernel's derivatives on the spot:
//             d/dz into per-lane registers (no cross-lane traffic for the whole run), d/dx into an LDS tile [time][feature][lane].
// Every TBT time steps the tile is summed over the 64 lanes (each thread adds 16 consecutive lanes of one row, a quad of threads
// finishes the row with two DPP moves) and leaves the chip ONCE per (role, tensor block, sequence, time, feature): no atomics, and --
// a workgroup being one wavefront -- no barrier anyone waits at; roles of different weight are balanced by the dispatcher instead of
// waiting for each other.  The partial sums are added up by tvs_grad_reduce_gx_kernel, the per-run partial d/dz by
// tvs_grad_reduce_gz_kernel.  (First form of this kernel, same round: the roles as wavefronts of one workgroup sharing the staged
// sequence and the tile -- two barriers per TBT steps and a 4 : 3 : 3 split made it 14.7 ms where this form takes less;
// profiles/r03_tvs_grad.txt.)  The kernel this replaces (tvs_grad_lanet_kernel, grad_kernels.hpp) swept time once per LEVEL and paid
// one LDS transpose + barrier + atomic per (level, sequence, time step).
//
// Base kernel at compile time: BASE_LINEAR, BASE_RBF (points prepared in units of sqrt(ln2/256): the inner product plus the two
// half squared norms is the argument of the table-driven 2^(t/256), fast_exp.hpp; gradients are scaled back by the reduction
// kernels), or -1: a run-time family through base_eval_grad (grad_core.hpp).
#pragma once

#include <type_traits>

#include "fast_exp.hpp"
#include "grad_core.hpp"
#include "tvs_plan.hpp"

namespace gpsig {

constexpr int TVSG_TBT = 2;              // time steps per flush of the d/dx tile (2 x D x 4 <= 64 items: one pass of the wave; a larger tile
                                         // costs LDS, i.e. the eighth workgroup of a CU, and no fewer instructions per step)
constexpr int TVSG_ROW = 66;             // doubles per tile row: 64 lanes + 2 (16-byte aligned rows, shifted by 4 banks each)
constexpr int TVSG_REC_ALIGN = 128;      // record granule in doubles (64 lanes x 16 bytes of LDS-DMA)

struct TvsGradTileArgs {
    const double* XR;    // (N, rec_elems) records: L rows of D prepared points (pre * x, zero beyond d), then L squared norms of those rows
    const double* ZL;    // (lt, E, D, Tpad) prepared components (pre * z), tensor index fastest
    const double* ZN;    // (lt, E, Tpad) their squared norms
    const double* Gt;    // fac == NULL: (N, M+1, Tpad) upstream gradient of the levels, tensor index fastest, zero beyond T;
                         // fac != NULL: (N, Tpad) upstream gradient of the weighted level sum  sum_m fac[n][m] level_m[t][n]
    const double* fac;   // (N, M+1) per-sequence level factors, or NULL
    double* gfp;         // fac != NULL: (tensor blocks, N, M+1) partial d/d fac of every tensor block
    const double* aux;   // (N, lt, Tpad) chain totals left by the forward tile kernel, or NULL: then this kernel sweeps forward itself
    double* gzp;         // (runs, lt, E, D, Tpad) partial d/dz' of every run
    double* gxp;         // (roles, tensor blocks, N, L, D) partial d/dx' of every (role, tensor block)
    double* gbp;         // (roles * runs * tensor blocks) partial d/d base_params[0], or NULL
    int64_t N, Tn, Tpad;
    int32_t L, d, kind, difference, M;
    int32_t run;         // sequences per workgroup
    int32_t rec_elems;   // multiple of TVSG_REC_ALIGN
    int32_t order;       // HO instances: min(order, num_levels) of the higher-order chains (signature_algs.py:129-160), <= TVSG_MAX_ORDER
    double p0, p1;
    double mat_a1, mat_a2, mat_g;   // TVSG_MATERN: P(u) = 1 + a1 u + a2 u^2 (Matern-1/2: 0, 0; 3/2: 1, 0; 5/2: 1, 1/3), g = pre * c
};
constexpr int TVSG_MAX_ORDER = 4;
// The three Matern families as ONE compile-time kind (round 6; what tvs_tile_kernel.hpp's forward instances did in round 5, here with wavefront-uniform
// coefficients instead of three instruction streams): points prepared in units of 1 / s, s = c 256 / ln2 (c = 1, sqrt 3, sqrt 5), so that with r' = s r the
// kernel is P(u) 2^(-r' / 256), u = c r = r' ln2 / 256 -- the table exp of the RBF instances, v_rsq_f64 + one Newton step for the root.  d kappa / dx =
// c (P' - P) 2^(-r'/256) (x' - z') / r' (kernels.py:955-993 differentiated); distances clamped at 1e-40 (:779-781) have derivative zero, as the reference's max().
constexpr int TVSG_MATERN = -2;
constexpr bool tvsg_has_table(int kind) { return kind == BASE_RBF || kind == TVSG_MATERN; }

// roles (level subsets, one workgroup each): at most four components per role where the levels allow it -- z, d/dz and the chain state
// of four components at six features fit the 256 registers of two wavefronts per SIMD; one wavefront per SIMD issues float64
// instructions at half the rate of two (tools/clockcheck.hip)
// (five components of the linear kernel, which carries no exp temporaries, still spill 100 bytes at six features: four for every family)
constexpr int tvs_grad_tile_roles(int M, int /*kind*/) {
    const int n = (M * (M + 1) / 2 + 3) / 4;
    return n > 4 ? 4 : n;
}

inline size_t tvs_grad_tile_lds_bytes(int D, int rec_elems, bool rbf) {
    return sizeof(double) * ((rbf ? EXP_TAB256_N : 0) + 2 * size_t(rec_elems) + size_t(TVSG_TBT) * D * TVSG_ROW);
}

__device__ __forceinline__ double tvsg_quad_xor1(double v) {        // lane ^ 1 within a quad
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true);  // quad_perm:[1,0,3,2]
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double tvsg_quad_xor2(double v) {        // lane ^ 2 within a quad
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, true);  // quad_perm:[2,3,0,1]
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// what the contraction of one component needs about kappa at one time point, for the point this lane holds
template <int KIND>
struct TvsgCoef {
    double k;            // kappa(z, x) of this lane's point
    double wz, vx, vz;   // d kappa/dx = wz z + vx x,  d kappa/dz = wz x + vz z      (generic families only; see TvsEv in grad_core.hpp)
    double dp0;
};
template <>
struct TvsgCoef<BASE_RBF> { double k; };       // wz = k, vx = vz = -k
template <>
struct TvsgCoef<BASE_LINEAR> { double k; };    // wz = 1, vx = vz = 0
template <>
struct TvsgCoef<TVSG_MATERN> { double k, vx; };     // d kappa/dx = vx (x - z), d kappa/dz = vx (z - x):  wz = -vx, vz = vx

// HO: the higher-order chains (signature_algs.py:129-160) at a run-time order.  Between time steps the state is the first-order one (the running
// totals U_j of chain j); within a step chain j splits by repeat count, r_j[0] = m_j U_{j-1}, r_j[l] = m_j r_{j-1}[l-1] / (l+1), U_j += sum_l r_j[l].
// The reverse step rebuilds the r of the step from the totals before it (ascending in j), then runs the adjoints down:
//   dL/dr_j[l] = W_j + m_{j+1} / (l+2) dL/dr_{j+1}[l+1],   dL/dm_j = dL/dr_j[0] U_{j-1} + sum_{l>=1} dL/dr_j[l] r_{j-1}[l-1] / (l+1),   W_{j-1} += m_j dL/dr_j[0]
// (wide_kernels.hpp: wide_chain_bwd holds the same step for the wide route).
template <int M, int D, int KIND, bool PAIRED, int MASK, bool HO = false>
struct TvsGradWave {
    static constexpr int NC = tvs_mask_comps(MASK);
    static constexpr int MASK_ = MASK;
    double z[NC][D];
    double zn[NC];          // RBF: -|z'|^2 / 2; otherwise |z|^2
    double gz[NC][D];       // sum over the run of (dL/dkappa * wz) x
    double bz[NC];          // sum over the run of  dL/dkappa * vz      (d/dz = gz + bz z)
    double u[NC], w[NC], gprev[NC];
    TvsgCoef<KIND> cn[NC];  // kappa (and derivative coefficients) at the LATER time point of the current increment
    double gp0;
    double sgn;             // PAIRED: -1 for the first point of a tensor, +1 for the second (kernels.py:329-330); unused otherwise

    __device__ __forceinline__ double signed_(double v) const {      // this lane's point enters kz with its sign
        if constexpr (PAIRED) return sgn * v;
        else return v;
    }

    __device__ __forceinline__ double combine(double v) const {     // sum over the two points of a tensor
        if constexpr (PAIRED) return v + tvsg_quad_xor1(v);
        else return v;
    }

    // lanes without a tensor (t >= Tn) get a harmless finite point; their upstream gradients are zero, so everything they add to the
    // d/dx sums is an exact zero (a zero point would make the cosine kernel's value, and with it 0 * NaN, a NaN)
    __device__ __forceinline__ void load(const TvsGradTileArgs& A, int64_t t, int e) {
        constexpr int E = PAIRED ? 2 : 1;
        const bool valid = t < A.Tn;
#pragma unroll
        for (int i = 1; i <= M; ++i) {
            if (!((MASK >> i) & 1)) continue;
#pragma unroll
            for (int j = 0; j < i; ++j) {
                const int c = tvs_local_off(MASK, i) + j, k = i * (i - 1) / 2 + j;
                const double s = valid ? A.ZN[(int64_t(k) * E + e) * A.Tpad + t] : double(D);
                zn[c] = KIND == BASE_RBF ? -0.5 * s : s;
#pragma unroll
                for (int f = 0; f < D; ++f) {
                    z[c][f] = valid ? A.ZL[((int64_t(k) * E + e) * D + f) * A.Tpad + t] : 1.0;
                    gz[c][f] = 0.0;
                }
                bz[c] = 0.0;
            }
        }
        gp0 = 0.0;
    }

    // the chain totals as the forward tile kernel left them (TvsTileArgs::aux) instead of a forward sweep
    __device__ __forceinline__ void load_totals(const TvsGradTileArgs& A, int64_t n, int64_t t) {
        constexpr int lt = M * (M + 1) / 2;
#pragma unroll
        for (int i = 1; i <= M; ++i) {
            if (!((MASK >> i) & 1)) continue;
#pragma unroll
            for (int j = 0; j < i; ++j) u[tvs_local_off(MASK, i) + j] = A.aux[(n * lt + i * (i - 1) / 2 + j) * A.Tpad + t];
        }
    }

    // kappa of this lane's point for every component at time tau; WITH_GRAD also leaves the derivative coefficients
    template <bool WITH_GRAD>
    __device__ __forceinline__ void eval(const TvsGradTileArgs& A, const double* __restrict__ rec, const double* __restrict__ etab, int tau,
                                         TvsgCoef<KIND> (&out)[NC]) const {
        double x[D];
#pragma unroll
        for (int f = 0; f < D; ++f) x[f] = rec[tau * D + f];
        const double xs = rec[A.L * D + tau];
        if constexpr (KIND == BASE_RBF) {
            const double hx = -0.5 * xs;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double t = zn[c] + hx;
#pragma unroll
                for (int f = 0; f < D; ++f) t = fma(z[c][f], x[f], t);
                out[c].k = kexp2_tab256(t, etab);
            }
        } else if constexpr (KIND == TVSG_MATERN) {
            constexpr double K = 0x1.62e42fefa39efp-1 / 256.0;                  // u = c r = r' ln2 / 256
            const double floor2 = 1e-40 * (A.mat_g * A.mat_g);                  // (>= the clamp of kernels.py:781 in prepared units; g = pre * c >= pre)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double ip = z[c][0] * x[0];
#pragma unroll
                for (int f = 1; f < D; ++f) ip = fma(z[c][f], x[f], ip);
                const double d2 = fma(-2.0, ip, zn[c] + xs);
                const bool clamped = !(d2 > floor2);
                const double dd = clamped ? floor2 : d2;
                const double y = __builtin_amdgcn_rsq(dd);
                double r = dd * y;
                r = fma(fma(-r, r, dd), 0.5 * y, r);                            // one Newton step on the residual
                const double e = kexp2_tab256(-r, etab);
                const double u = r * K;
                const double pu = fma(fma(A.mat_a2, u, A.mat_a1), u, 1.0);
                out[c].k = pu * e;
                if (WITH_GRAD) {
                    const double dpu = fma(2.0 * A.mat_a2, u, A.mat_a1);
                    out[c].vx = clamped ? 0.0 : (A.mat_g * (dpu - pu) * e) / r;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double ip = z[c][0] * x[0];
#pragma unroll
                for (int f = 1; f < D; ++f) ip = fma(z[c][f], x[f], ip);
                if constexpr (KIND == BASE_LINEAR) {
                    out[c].k = ip;
                } else if (WITH_GRAD) {
                    const BaseGrad g = base_eval_grad(A.kind, ip, zn[c], xs, A.p0, A.p1);      // first argument z, second x
                    out[c].k = g.k; out[c].wz = g.cy - g.cd; out[c].vx = g.cx2 + g.cd; out[c].vz = g.cx + g.cd; out[c].dp0 = g.dp0;
                } else {
                    out[c].k = base_eval<double>(A.kind, ip, zn[c], xs, A.p0, A.p1);
                }
            }
        }
    }

    // forward chains of this wave's levels over one sequence (signature_algs.py:114-125): leaves u = the chain totals
    __device__ __forceinline__ void forward(const TvsGradTileArgs& A, const double* __restrict__ rec, const double* __restrict__ etab) {
#pragma unroll
        for (int c = 0; c < NC; ++c) u[c] = 0.0;
        const int L = A.L;
        TvsgCoef<KIND> ka[NC], kb[NC];
        auto chains = [&](const TvsgCoef<KIND> (&hi)[NC], const TvsgCoef<KIND> (&lo)[NC], bool diff) {
            double dm[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) dm[c] = combine(signed_(diff ? hi[c].k - lo[c].k : hi[c].k));
#pragma unroll
            for (int i = 1; i <= M; ++i) {
                if (!((MASK >> i) & 1)) continue;
                const int c0 = tvs_local_off(MASK, i);
                if constexpr (HO) {
                    if (i > 1) {
                        constexpr int O = TVSG_MAX_ORDER;
                        double rp[O], uold = u[c0];
                        rp[0] = dm[c0];
                        u[c0] += dm[c0];
#pragma unroll
                        for (int j = 1; j < i; ++j) {
                            double rc[O], tot = dm[c0 + j] * uold;
                            rc[0] = tot;
#pragma unroll
                            for (int l = 1; l < O; ++l) {
                                rc[l] = (l <= j && l < A.order) ? (dm[c0 + j] * (1.0 / double(l + 1))) * rp[l - 1] : 0.0;
                                tot += rc[l];
                            }
                            uold = u[c0 + j];
                            u[c0 + j] += tot;
#pragma unroll
                            for (int l = 0; l < O; ++l) rp[l] = rc[l];
                        }
                        continue;
                    }
                }
#pragma unroll
                for (int j = i - 1; j >= 1; --j) u[c0 + j] = fma(dm[c0 + j], u[c0 + j - 1], u[c0 + j]);
                u[c0] += dm[c0];
            }
        };
        if (!A.difference) {
            for (int tau = 0; tau < L; ++tau) {
                eval<false>(A, rec, etab, tau, ka);
                chains(ka, ka, false);
            }
        } else {
            eval<false>(A, rec, etab, 0, ka);
            int tau = 1;
            for (; tau + 1 < L; tau += 2) {                       // two steps per trip: the previous values alternate registers
                eval<false>(A, rec, etab, tau, kb);
                chains(kb, ka, true);
                eval<false>(A, rec, etab, tau + 1, ka);
                chains(ka, kb, true);
            }
            if (tau < L) {
                eval<false>(A, rec, etab, tau, kb);
                chains(kb, ka, true);
            }
        }
    }

    // contraction at one time point: gk[c] = dL/d kz_c(x_time) (the same in both lanes of a pair), co = the coefficients there
    template <class Emit>
    __device__ __forceinline__ void contract(const double* __restrict__ rec, int time, const double (&gk)[NC], const TvsgCoef<KIND> (&co)[NC],
                                             Emit&& emit) {
        double x[D], gx[D];
#pragma unroll
        for (int f = 0; f < D; ++f) { x[f] = rec[time * D + f]; gx[f] = 0.0; }
        double sbx = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double g = signed_(gk[c]);
            double a;
            if constexpr (KIND == BASE_LINEAR) {
                a = g;
            } else if constexpr (KIND == BASE_RBF) {
                a = g * co[c].k;
                sbx -= a;
                bz[c] -= a;
            } else if constexpr (KIND == TVSG_MATERN) {
                const double gv = g * co[c].vx;
                a = -gv;
                sbx += gv;
                bz[c] += gv;
            } else {
                a = g * co[c].wz;
                sbx = fma(g, co[c].vx, sbx);
                bz[c] = fma(g, co[c].vz, bz[c]);
                gp0 = fma(g, co[c].dp0, gp0);
            }
#pragma unroll
            for (int f = 0; f < D; ++f) {
                gx[f] = fma(a, z[c][f], gx[f]);
                gz[c][f] = fma(a, x[f], gz[c][f]);
            }
        }
        if constexpr (KIND != BASE_LINEAR) {
#pragma unroll
            for (int f = 0; f < D; ++f) gx[f] = fma(sbx, x[f], gx[f]);
        }
        emit(time, gx);
    }

    // one reverse step of level i's higher-order chains (components c0 .. c0 + i - 1): u back to the totals before the step, gm = dL/dm, w updated
    __device__ __forceinline__ void undo_ho(const int i, const int c0, const double (&m)[NC], double (&gm)[NC], double up, int order) {
        constexpr int O = TVSG_MAX_ORDER;
        double r[M][O], ub[M];
        ub[0] = u[c0] - m[c0];
#pragma unroll
        for (int l = 0; l < O; ++l) r[0][l] = l == 0 ? m[c0] : 0.0;
#pragma unroll
        for (int j = 1; j < M; ++j) {
            if (j >= i) break;
            double tot = m[c0 + j] * ub[j - 1];
            r[j][0] = tot;
#pragma unroll
            for (int l = 1; l < O; ++l) {
                r[j][l] = (l <= j && l < order) ? (m[c0 + j] * (1.0 / double(l + 1))) * r[j - 1][l - 1] : 0.0;
                tot += r[j][l];
            }
            ub[j] = u[c0 + j] - tot;
        }
        double gn[O], add[M];
#pragma unroll
        for (int l = 0; l < O; ++l) gn[l] = 0.0;
#pragma unroll
        for (int j = M - 1; j >= 0; --j) {
            if (j >= i) continue;
            const double wj = (j == i - 1) ? up : w[c0 + (j + 1 < i ? j + 1 : j)];
            double gr[O];
#pragma unroll
            for (int l = 0; l < O; ++l) {
                const bool live = l <= j && l < order;
                const bool upl = j + 1 < i && l + 1 < O && l + 1 < order;
                gr[l] = live ? wj + (upl ? (m[c0 + (j + 1 < i ? j + 1 : j)] * (1.0 / double(l + 2))) * gn[l + 1 < O ? l + 1 : l] : 0.0) : 0.0;
            }
            double gd = gr[0] * (j >= 1 ? ub[j >= 1 ? j - 1 : 0] : 1.0);
#pragma unroll
            for (int l = 1; l < O; ++l)
                if (j >= 1) gd = fma(gr[l] * (1.0 / double(l + 1)), r[j >= 1 ? j - 1 : 0][l - 1], gd);
            gm[c0 + j] = gd;
            add[j] = m[c0 + j] * gr[0];
#pragma unroll
            for (int l = 0; l < O; ++l) gn[l] = gr[l];
        }
#pragma unroll
        for (int j = 1; j < M; ++j)
            if (j < i) w[c0 + j] += add[j];
#pragma unroll
        for (int j = 0; j < M; ++j)
            if (j < i) u[c0 + j] = ub[j];
    }

    // backward sweep over one sequence; cup[i] = upstream gradient of level i for this (tensor, sequence)
    template <class Emit>
    __device__ __forceinline__ void backward(const TvsGradTileArgs& A, const double* __restrict__ rec, const double* __restrict__ etab,
                                             const double (&cup)[M + 1], Emit&& emit) {
        const int L = A.L;
        const bool diff = A.difference != 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) { w[c] = 0.0; gprev[c] = 0.0; }
        if (diff) eval<true>(A, rec, etab, L - 1, cn);
        for (int tau = (diff ? L - 2 : L - 1); tau >= 0; --tau) {
            TvsgCoef<KIND> cc[NC];
            eval<true>(A, rec, etab, tau, cc);
            double m[NC], gm[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) m[c] = combine(signed_(diff ? cn[c].k - cc[c].k : cc[c].k));
#pragma unroll
            for (int i = 1; i <= M; ++i) {
                if (!((MASK >> i) & 1)) continue;
                const int c0 = tvs_local_off(MASK, i);
                if constexpr (HO) {
                    if (i > 1) {
                        undo_ho(i, c0, m, gm, cup[i], A.order);
                        continue;
                    }
                }
                // undo, lowest chain first: afterwards u[c0 + j] = u_{j+1}[tau-1]; `below` = u_j[tau-1], what m_j[tau] was multiplied with
                double below = 1.0;
#pragma unroll
                for (int j = 0; j < i; ++j) {
                    const double wnext = (j == i - 1) ? cup[i] : w[c0 + j + 1];       // W_{j+1}[tau], before this step's contribution
                    gm[c0 + j] = below * wnext;
                    const double ub = below;
                    u[c0 + j] = fma(-m[c0 + j], ub, u[c0 + j]);
                    below = u[c0 + j];
                    if (j >= 1) w[c0 + j] = fma(m[c0 + j], wnext, w[c0 + j]);        // W_j[tau-1] += m_j[tau] W_{j+1}[tau]
                }
            }
            if (diff) {
                double gk[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) { gk[c] = gm[c] - gprev[c]; gprev[c] = gm[c]; }      // dL/d kz(x_{tau+1})
                contract(rec, tau + 1, gk, cn, emit);
#pragma unroll
                for (int c = 0; c < NC; ++c) cn[c] = cc[c];
            } else {
                contract(rec, tau, gm, cc, emit);
            }
        }
        if (diff) {                                                   // time point 0
            double gk[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) gk[c] = -gprev[c];
            contract(rec, 0, gk, cn, emit);
        }
    }
};

#ifndef TVSG_WAVES_PER_EU
#define TVSG_WAVES_PER_EU 2
#endif
// grid (tensor blocks, runs, roles); block 64
template <int M, int D, int KIND, bool PAIRED, bool HO = false>
__global__ __launch_bounds__(64, TVSG_WAVES_PER_EU) void tvs_grad_tile_kernel(const TvsGradTileArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tvsg_smem[];
    constexpr int NR = tvs_grad_tile_roles(M, KIND);
    constexpr int NTAB = tvsg_has_table(KIND) ? EXP_TAB256_N : 0;
    constexpr int TPW = PAIRED ? 32 : 64;                          // tensors per workgroup
    constexpr int E = PAIRED ? 2 : 1;
    double* const etab = reinterpret_cast<double*>(tvsg_smem);
    double* const recs = etab + NTAB;                              // 2 x rec_elems
    double* const gxt = recs + 2 * A.rec_elems;                    // [TBT][D][TVSG_ROW]
    const int lane = threadIdx.x;
    const int role = blockIdx.z;
    const int pe = PAIRED ? (lane & 1) : 0;
    const int64_t t = blockIdx.x * int64_t(TPW) + (PAIRED ? lane >> 1 : lane);      // < Tpad
    const int64_t n_begin = blockIdx.y * int64_t(A.run);
    const int64_t n_end = (n_begin + A.run < A.N) ? n_begin + A.run : A.N;
    const int lt = M * (M + 1) / 2;
    double* const gxp = A.gxp + (int64_t(role) * gridDim.x + blockIdx.x) * A.N * A.L * D;

    if constexpr (tvsg_has_table(KIND)) exp_tab256_fill(etab, lane, 64);

    auto stage = [&](int64_t n, int buf) {
        const double* src = A.XR + n * int64_t(A.rec_elems);
        double* dst = recs + buf * A.rec_elems;
        for (int c = 0; c < A.rec_elems; c += 128)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c + lane * 2),
                                             (__attribute__((address_space(3))) void*)(dst + c), 16, 0, 0);
    };
    if (n_begin < n_end) stage(n_begin, 0);

    auto run_role = [&](auto& W) {
        using WT = typename std::remove_reference<decltype(W)>::type;
        W.sgn = (PAIRED && pe == 0) ? -1.0 : 1.0;
        W.load(A, t, pe);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                          // one wavefront: orders the LDS-DMA and the table against the reads below
        for (int64_t n = n_begin; n < n_end; ++n) {
            const int buf = int(n - n_begin) & 1;
            if (n + 1 < n_end) stage(n + 1, buf ^ 1);
            double cup[M + 1];
            double gsum = 0.0;                                    // weighted mode: upstream gradient of the level sum at (t, n)
            if (A.fac) {
                gsum = A.Gt[n * A.Tpad + t];
#pragma unroll
                for (int i = 0; i <= M; ++i) cup[i] = ((WT::MASK_ >> i) & 1) ? gsum * A.fac[n * (M + 1) + i] : 0.0;
            } else {
#pragma unroll
                for (int i = 0; i <= M; ++i) cup[i] = ((WT::MASK_ >> i) & 1) ? A.Gt[(n * (M + 1) + i) * A.Tpad + t] : 0.0;
            }
            const double* rec = recs + buf * A.rec_elems;
            if (A.aux) W.load_totals(A, n, t);
            else W.forward(A, rec, etab);
            if (A.fac) {
                // d/d fac[n][i] = sum over the tensors of (upstream gradient) x (level value): the chain totals are at hand
                // (lanes without a tensor carry a zero gradient; the two lanes of an incremental tensor hold the same numbers)
                const double gq = PAIRED ? 0.5 * gsum : gsum;
                if (role == 0) {                                  // level 0 == 1 (signature_algs.py:116)
                    const double s0 = grad_wave_sum(gq);
                    if (lane == 0) A.gfp[(blockIdx.x * A.N + n) * (M + 1)] = s0;
                }
#pragma unroll
                for (int i = 1; i <= M; ++i) {
                    if (!((WT::MASK_ >> i) & 1)) continue;
                    const double si = grad_wave_sum(gq * W.u[tvs_local_off(WT::MASK_, i) + i - 1]);
                    if (lane == 0) A.gfp[(blockIdx.x * A.N + n) * (M + 1) + i] = si;
                }
            }
            // d/dx of this sequence: the lanes' partial sums go to the tile, every TBT time steps the wave adds them up
            auto emit = [&](int time, const double (&gx)[D]) {
                const int tb = time % TVSG_TBT;
#pragma unroll
                for (int f = 0; f < D; ++f) gxt[(tb * D + f) * TVSG_ROW + lane] = gx[f];
                if (tb == 0) {
                    __syncthreads();
                    for (int item = lane; item < TVSG_TBT * D * 4; item += 64) {
                        const int r = item >> 2, q = item & 3;            // row (time in batch, feature), quarter of the lanes
                        const double2* row = reinterpret_cast<const double2*>(gxt + r * TVSG_ROW + 16 * q);
                        double s0 = 0.0, s1 = 0.0;
#pragma unroll
                        for (int e2 = 0; e2 < 8; e2 += 2) {
                            const double2 v0 = row[e2], v1 = row[e2 + 1];
                            s0 += v0.x + v0.y;
                            s1 += v1.x + v1.y;
                        }
                        double s = s0 + s1;
                        s += tvsg_quad_xor1(s);
                        s += tvsg_quad_xor2(s);
                        const int rb = r / D, f = r - rb * D;
                        if (q == 0 && time + rb < A.L) gxp[(n * A.L + time + rb) * D + f] = s;
                    }
                    __syncthreads();
                }
            };
            W.backward(A, rec, etab, cup, emit);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next record has landed
            __syncthreads();
        }
        // d/dz' of the run: gz + bz z, one partial per (run, component, point, feature), tensor index fastest
#pragma unroll
        for (int i = 1; i <= M; ++i) {
            if (!((WT::MASK_ >> i) & 1)) continue;
#pragma unroll
            for (int j = 0; j < i; ++j) {
                const int c = tvs_local_off(WT::MASK_, i) + j, k = i * (i - 1) / 2 + j;
#pragma unroll
                for (int f = 0; f < D; ++f)
                    A.gzp[(((blockIdx.y * int64_t(lt) + k) * E + pe) * D + f) * A.Tpad + t] =
                        KIND == BASE_LINEAR ? W.gz[c][f] : fma(W.bz[c], W.z[c][f], W.gz[c][f]);     // the linear kernel has no z term
            }
        }
        if (A.gbp) {
            const double s = grad_wave_sum(W.gp0);
            if (lane == 0) A.gbp[(int64_t(role) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
        }
    };

#define TVSG_ROLE(r_)                                                          \
    {                                                                          \
        TvsGradWave<M, D, KIND, PAIRED, tvs_level_mask(M, NR, r_), HO> W;         \
        run_role(W);                                                           \
    }
    if constexpr (NR == 1) {
        TVSG_ROLE(0)
    } else if constexpr (NR == 2) {
        if (role == 0) TVSG_ROLE(0) else TVSG_ROLE(1)
    } else if constexpr (NR == 3) {
        if (role == 0) TVSG_ROLE(0) else if (role == 1) TVSG_ROLE(1) else TVSG_ROLE(2)
    } else {
        if (role == 0) TVSG_ROLE(0) else if (role == 1) TVSG_ROLE(1) else if (role == 2) TVSG_ROLE(2) else TVSG_ROLE(3)
    }
#undef TVSG_ROLE
}

// G (M+1, T, N) -> Gt (N, M+1, Tpad), zero for t >= T: a lane (= tensor) of the tile kernel then reads its upstream gradients from
// consecutive addresses.  grid (ceil(N / 32), Tpad / 32, M+1), block (32, 8).
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void tvs_grad_transpose_G_kernel(const double* __restrict__ G, int64_t Tn, int64_t Tpad, int64_t N, int M1,
                                                   double* __restrict__ Gt) {
    __shared__ double tile[32][33];
    const int lv = blockIdx.z;
    const int64_t n0 = blockIdx.x * int64_t(32), t0 = blockIdx.y * int64_t(32);
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int64_t tt = t0 + r, n = n0 + threadIdx.x;
        tile[r][threadIdx.x] = (tt < Tn && n < N) ? G[(int64_t(lv) * Tn + tt) * N + n] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int64_t n = n0 + r, tt = t0 + threadIdx.x;
        if (n < N) Gt[(n * M1 + lv) * Tpad + tt] = tile[threadIdx.x][r];
    }
}
#else
__global__ void tvs_grad_transpose_G_kernel(const double* __restrict__ G, int64_t Tn, int64_t Tpad, int64_t N, int M1,
                                                   double* __restrict__ Gt);
#endif

// gX[n][tau][f] = scale * sum over the (role, tensor block) partials of gxp[b][n][tau][f]      (D-wide rows -> the caller's d columns)
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void tvs_grad_reduce_gx_kernel(const double* __restrict__ gxp, int nblocks, int64_t rows /* N * L */, int D, int d,
                                                 double scale, double* __restrict__ gX) {
    const int64_t total = rows * d;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += int64_t(gridDim.x) * blockDim.x) {
        const int64_t row = e / d;
        const int f = int(e - row * d);
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += gxp[(int64_t(b) * rows + row) * D + f];
        gX[e] = s * scale;
    }
}
#else
__global__ void tvs_grad_reduce_gx_kernel(const double* __restrict__ gxp, int nblocks, int64_t rows /* N * L */, int D, int d,
                                                 double scale, double* __restrict__ gX);
#endif

// gfac[n][i] = sum over the tensor blocks of gfp[b][n][i]
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void tvs_grad_reduce_gf_kernel(const double* __restrict__ gfp, int nblocks, int64_t n, double* __restrict__ gfac) {
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += int64_t(gridDim.x) * blockDim.x) {
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += gfp[int64_t(b) * n + e];
        gfac[e] = s;
    }
}
#else
__global__ void tvs_grad_reduce_gf_kernel(const double* __restrict__ gfp, int nblocks, int64_t n, double* __restrict__ gfac);
#endif

// gZ[k][t][e][f] = scale * sum over the runs of gzp[run][k][e][f][t].  collapse (linear kernel with incremental tensors: the
// kernel saw z1 - z0): the caller's two points receive (-g, +g).
#ifdef GPSIG_KERNEL_DEFS          // defined once, in kernel_defs.hip; every other unit sees the declaration
__global__ void tvs_grad_reduce_gz_kernel(const double* __restrict__ gzp, int nruns, int lt, int E, int D, int64_t Tpad, int64_t Tn,
                                                 int d, int collapse, double scale, double* __restrict__ gZ) {
    const int64_t total = int64_t(lt) * E * d * Tn;
    for (int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * blockDim.x) {
        const int64_t t = idx % Tn;
        const int64_t kef = idx / Tn;
        const int f = int(kef % d);
        const int e = int((kef / d) % E);
        const int k = int(kef / (int64_t(d) * E));
        const int64_t stride = int64_t(lt) * E * D * Tpad;
        const double* src = gzp + ((int64_t(k) * E + e) * D + f) * Tpad + t;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int r = 0;
        for (; r + 4 <= nruns; r += 4) {
            s0 += src[int64_t(r) * stride];
            s1 += src[int64_t(r + 1) * stride];
            s2 += src[int64_t(r + 2) * stride];
            s3 += src[int64_t(r + 3) * stride];
        }
        for (; r < nruns; ++r) s0 += src[int64_t(r) * stride];
        const double g = ((s0 + s1) + (s2 + s3)) * scale;
        if (collapse) {
            gZ[((int64_t(k) * Tn + t) * 2 + 0) * d + f] = -g;
            gZ[((int64_t(k) * Tn + t) * 2 + 1) * d + f] = g;
        } else {
            gZ[((int64_t(k) * Tn + t) * E + e) * d + f] = g;
        }
    }
}
#else
__global__ void tvs_grad_reduce_gz_kernel(const double* __restrict__ gzp, int nruns, int lt, int E, int D, int64_t Tpad, int64_t Tn,
                                                 int d, int collapse, double scale, double* __restrict__ gZ);
#endif

}  // namespace gpsig
