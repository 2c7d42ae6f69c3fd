// grad_core.hpp -- reverse-mode derivatives of the signature-kernel recursions, one (x, y) pair per thread.
//
// The reference has no gradient code: it is trained by TensorFlow's automatic differentiation of the graph that
// gpsig/signature_algs.py builds (training.py:149-164, models.py:40-59).  What autodiff computes for that graph is
// restated here in closed form.  Shared verbatim by the gfx950 kernels (grad_kernels.hpp) and by the host-side test
// harness under tests/, which runs the same per-pair code with plain adds instead of atomics.
//
// Sequence vs sequence, first-order algorithm (signature_algs.py:8-35).  With the increment lattice dM[a][b]
// (a < R1, b < R2), Q_m the inclusive 2-D prefix of R_m (Q_0 == 1) and R_m = dM * Q_{m-1}[a-1][b-1]:
//     K_m = sum R_m,    loss L = sum_m c_m K_m     (c_m = upstream gradient of level m for this pair)
//     U_p[a][b] := dL/dR_p[a][b] = c_p + sum_{a'>a, b'>b} dM[a'][b'] U_{p+1}[a'][b']        (U_M == c_M)
//     Lam[a][b] := dL/ddM[a][b]  = sum_{p=1..M} Q_{p-1}[a-1][b-1] * U_p[a][b]
// Phase F sweeps the lattice forward and keeps every Q_m[a][b] (m < M) and dM[a][b]; phase B sweeps it backward,
// overwrites Q_m by the suffix sums of dM*U_{m+1} in place and dM by Lam; phase C contracts Lam with the base
// kernel's derivatives into gradients with respect to the (already scaled) observations of both sequences.
// The lattice lives in a scratch array laid out [slot][a][b][pair] so that the 64 pairs of a wavefront touch 64
// consecutive doubles; slot 0 = dM / Lam, slot m = Q_m.
//
// Tensor vs sequence, first-order algorithm (signature_algs.py:101-127): the same idea on 1-D chains, see TvsGrad.
#pragma once

#include "seq_core.hpp"

namespace gpsig {

// d kappa / d x[f] = cy * y[f] + cx * x[f] + cd * (x[f] - y[f]),   d kappa / d y[f] = cy * x[f] + cx2 * y[f] - cd * (x[f] - y[f]),
// dp0 = d kappa / d base_params[0]  (gamma of SignaturePoly :844-848, mixing of SignatureMix :881-892).
// Derivatives as TensorFlow's autodiff takes them of the reference's formulas (gpsig/kernels.py:765-781, 799-993); in
// particular sqrt(max(r2, 1e-40)) (:779-781) has zero derivative where the clamp is active.
struct BaseGrad {
    double k, cy, cx, cx2, cd, dp0;
};

GPSIG_HD BaseGrad base_eval_grad(int kind, double inner, double xs, double ys, double p0, double p1) {
    BaseGrad g;
    g.cy = g.cx = g.cx2 = g.cd = g.dp0 = 0.0;
    switch (kind) {
        case BASE_LINEAR: g.k = inner; g.cy = 1.0; return g;
        case BASE_COSINE: {
            const double sx = sqrt(xs), sy = sqrt(ys);
            g.k = inner / (sx * sy);
            g.cy = 1.0 / (sx * sy);
            g.cx = -g.k / xs;        // d/dx of 1/sqrt(xs) = -x / xs^{3/2}
            g.cx2 = -g.k / ys;
            return g;
        }
        case BASE_POLY: {
            const double b = inner + p0;
            const double bm = poly_pow(b, p1 - 1.0);     // (seq_core.hpp: a whole exponent by repeated squaring)
            g.k = (double(int(p1)) == p1 && p1 >= 1.0 && p1 <= 9.0) ? bm * b : pow(b, p1);
            g.cy = p1 * bm;
            g.dp0 = g.cy;
            return g;
        }
        default: break;
    }
    const double dist = fma(-2.0, inner, xs + ys);
    if (kind == BASE_RBF) {
        g.k = exp(-dist / 2);
        g.cd = -g.k;                 // d kappa/d dist = -kappa/2,  d dist/dx = 2 (x - y)
        return g;
    }
    if (kind == BASE_MIX) {
        const double e = exp(-dist / 2);
        g.k = p0 * e + (1.0 - p0) * inner;
        g.cd = -p0 * e;
        g.cy = 1.0 - p0;
        g.dp0 = e - inner;
        return g;
    }
    const bool clamped = !(dist > 1e-40);
    const double r = sqrt(fmax(dist, 1e-40));
    double dk_dr;
    if (kind == BASE_MATERN12) {
        g.k = exp(-r);
        dk_dr = -g.k;
    } else if (kind == BASE_MATERN32) {
        const double c = 1.7320508075688772935, e = exp(-c * r);
        g.k = (1.0 + c * r) * e;
        dk_dr = -3.0 * r * e;
    } else {
        const double c = 2.2360679774997896964, e = exp(-c * r);
        g.k = (1.0 + c * r + (5.0 / 3.0) * (r * r)) * e;
        dk_dr = -(5.0 / 3.0) * r * (1.0 + c * r) * e;
    }
    g.cd = clamped ? 0.0 : dk_dr / r;    // dk/ddist = dk_dr / (2 r);  d dist/dx = 2 (x - y)
    return g;
}

#if defined(__HIPCC__)
__device__ __forceinline__ double grad_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// all lanes of the wavefront call this together; `uniform` = every lane targets the same address
__device__ __forceinline__ void grad_add(double* p, double v, bool uniform, bool valid) {
    if (uniform) {
        v = grad_wave_sum(valid ? v : 0.0);
        if ((threadIdx.x & 63) == 0) atomicAdd(p, v);
    } else if (valid) {
        atomicAdd(p, v);
    }
}
#endif
// host-side stand-in (the test harness): a plain add
inline void grad_add(double* p, double v, bool, bool valid) { if (valid) *p += v; }

enum : int { GRAD_MAX_LEVELS = 8 };

// One launch of the sequence-pair gradient.  All arrays are float64.  Time-major point arrays: element (t, f) of
// sequence i is at  T[(t * DP + f) * stride + i]  (features padded with zeros to DP).
struct SeqGradArgs {
    const double* xT; const double* yT;     // scaled observations of the two sides
    double* gxT; double* gyT;               // gradients with respect to them (accumulated), same layout
    int64_t xstride, ystride;
    int N1, N2;                             // sequences on either side
    int L1, L2;                             // observations
    int M, kind, mode;                      // levels, base kernel, MODE_*
    double p0, p1;
    int diag;                               // 1: pairs (i, i), x and y are the same array (N2 == N1)
    int j0, nj;                             // this launch covers y sequences j0 .. j0+nj-1 (ignored when diag)
    const double* G;                        // upstream gradient of the level arrays: G[m * gm + i * gi + j * gj], m = 0..M
    int64_t gm, gi, gj;
    double* scratch;                        // M * R1 * R2 * pairs doubles
    int64_t pairs;                          // pairs of this launch = scratch stride of one lattice cell
    double* levels;                         // optional (M+1) x pairs: the forward levels recomputed on the way (tests)
    double* gbase;                          // optional 2 doubles: gradient with respect to base_params[0..1] (only [0] is produced)
};

template <int DP>
struct SeqPairGrad {
    const SeqGradArgs& A;
    int i, j;
    int64_t pidx;
    bool valid;
    int R1, R2;

    GPSIG_HD SeqPairGrad(const SeqGradArgs& a, int i_, int j_, int64_t pidx_, bool valid_) : A(a), i(i_), j(j_), pidx(pidx_), valid(valid_) {
        const int dr = A.mode == MODE_PT_NODIFF ? 0 : 1;
        R1 = A.L1 - dr;
        R2 = A.L2 - dr;
    }
    GPSIG_HD double& cell(int s, int a, int b) const { return A.scratch[((int64_t(s) * R1 + a) * R2 + b) * A.pairs + pidx]; }
    GPSIG_HD void load_x(int t, double (&v)[DP]) const {
#pragma unroll
        for (int f = 0; f < DP; ++f) v[f] = A.xT[(int64_t(t) * DP + f) * A.xstride + i];
    }
    GPSIG_HD void load_y(int t, double (&v)[DP]) const {
#pragma unroll
        for (int f = 0; f < DP; ++f) v[f] = A.yT[(int64_t(t) * DP + f) * A.ystride + j];
    }
    static GPSIG_HD double dot(const double (&a)[DP], const double (&b)[DP]) {
        double s = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f) s = fma(a[f], b[f], s);
        return s;
    }

    // ---- phase F: forward sweep, keeps dM and Q_1..Q_{M-1} for every lattice cell ----------------------------
    GPSIG_HD void forward() const {
        const int M = A.M;
        double ktop = 0.0, qlast[GRAD_MAX_LEVELS];
#pragma unroll
        for (int m = 0; m < GRAD_MAX_LEVELS; ++m) qlast[m] = 0.0;
        double x0[DP], x1[DP];
        if (A.mode != MODE_PT_NODIFF) load_x(0, x1);
        for (int a = 0; a < R1; ++a) {
            double xs0 = 0.0, xs1 = 0.0;
            if (A.mode == MODE_PT_NODIFF) {
                load_x(a, x0);
                xs0 = dot(x0, x0);
            } else {
#pragma unroll
                for (int f = 0; f < DP; ++f) x0[f] = x1[f];
                load_x(a + 1, x1);
                if (A.mode == MODE_INC) {
#pragma unroll
                    for (int f = 0; f < DP; ++f) x0[f] = x1[f] - x0[f];      // x0 := increment a
                } else {
                    xs0 = dot(x0, x0);
                    xs1 = dot(x1, x1);
                }
            }
            double s[GRAD_MAX_LEVELS + 1], qd[GRAD_MAX_LEVELS];
#pragma unroll
            for (int m = 0; m <= GRAD_MAX_LEVELS; ++m) s[m] = 0.0;
#pragma unroll
            for (int m = 0; m < GRAD_MAX_LEVELS; ++m) qd[m] = 0.0;
            double y0[DP], y1[DP], klo = 0.0, khi = 0.0;      // kappa(x_a, y_b), kappa(x_{a+1}, y_b)
            if (A.mode != MODE_PT_NODIFF) {
                load_y(0, y1);
                if (A.mode == MODE_PT_DIFF) {
                    const double ys = dot(y1, y1);
                    klo = base_eval<double>(A.kind, dot(x0, y1), xs0, ys, A.p0, A.p1);
                    khi = base_eval<double>(A.kind, dot(x1, y1), xs1, ys, A.p0, A.p1);
                }
            }
            for (int b = 0; b < R2; ++b) {
                double dm;
                if (A.mode == MODE_PT_NODIFF) {
                    load_y(b, y0);
                    dm = base_eval<double>(A.kind, dot(x0, y0), xs0, dot(y0, y0), A.p0, A.p1);
                } else {
#pragma unroll
                    for (int f = 0; f < DP; ++f) y0[f] = y1[f];
                    load_y(b + 1, y1);
                    if (A.mode == MODE_INC) {
                        double acc = 0.0;
#pragma unroll
                        for (int f = 0; f < DP; ++f) acc = fma(x0[f], y1[f] - y0[f], acc);
                        dm = acc;
                    } else {
                        const double ys = dot(y1, y1);
                        const double nlo = base_eval<double>(A.kind, dot(x0, y1), xs0, ys, A.p0, A.p1);
                        const double nhi = base_eval<double>(A.kind, dot(x1, y1), xs1, ys, A.p0, A.p1);
                        dm = (nhi - khi) - (nlo - klo);
                        klo = nlo;
                        khi = nhi;
                    }
                }
                cell(0, a, b) = dm;
                double qup[GRAD_MAX_LEVELS];
#pragma unroll
                for (int m = 1; m < GRAD_MAX_LEVELS; ++m) qup[m] = (m < M && a > 0) ? cell(m, a - 1, b) : 0.0;
                s[1] += dm;
#pragma unroll
                for (int m = 2; m <= GRAD_MAX_LEVELS; ++m)
                    if (m <= M) s[m] = fma(dm, qd[m - 1], s[m]);
#pragma unroll
                for (int m = 1; m < GRAD_MAX_LEVELS; ++m)
                    if (m < M) {
                        const double q = qup[m] + s[m];
                        cell(m, a, b) = q;
                        qd[m] = qup[m];
                        qlast[m] = q;
                    }
            }
#pragma unroll
            for (int m = 1; m <= GRAD_MAX_LEVELS; ++m)
                if (m == M) ktop += s[m];
        }
        if (A.levels && valid) {
            A.levels[pidx] = 1.0;
#pragma unroll
            for (int m = 1; m < GRAD_MAX_LEVELS; ++m)
                if (m < M) A.levels[int64_t(m) * A.pairs + pidx] = (R1 > 0 && R2 > 0) ? qlast[m] : 0.0;
            A.levels[int64_t(M) * A.pairs + pidx] = ktop;
        }
    }

    // ---- phase B: backward sweep; slot 0 becomes Lam, slot m the suffix sums of dM * U_{m+1} ------------------
    GPSIG_HD void backward() const {
        const int M = A.M;
        double c[GRAD_MAX_LEVELS + 1];
#pragma unroll
        for (int m = 0; m <= GRAD_MAX_LEVELS; ++m) c[m] = (m >= 1 && m <= M && valid) ? A.G[m * A.gm + i * A.gi + j * A.gj] : 0.0;
        for (int a = R1 - 1; a >= 0; --a) {
            double sv[GRAD_MAX_LEVELS], qbd[GRAD_MAX_LEVELS];
#pragma unroll
            for (int p = 0; p < GRAD_MAX_LEVELS; ++p) sv[p] = qbd[p] = 0.0;
            for (int b = R2 - 1; b >= 0; --b) {
                const double dm = cell(0, a, b);
                double U[GRAD_MAX_LEVELS + 2];
                double lam;
#pragma unroll
                for (int p = 1; p <= GRAD_MAX_LEVELS; ++p) U[p] = p < M ? c[p] + qbd[p] : (p == M ? c[p] : 0.0);
                U[GRAD_MAX_LEVELS + 1] = 0.0;
                lam = U[1];
#pragma unroll
                for (int p = 2; p <= GRAD_MAX_LEVELS; ++p)
                    if (p <= M && a > 0 && b > 0) lam = fma(cell(p - 1, a - 1, b - 1), U[p], lam);
                cell(0, a, b) = lam;
#pragma unroll
                for (int p = 1; p < GRAD_MAX_LEVELS; ++p)
                    if (p < M) {
                        const double qdn = (a + 1 < R1) ? cell(p, a + 1, b) : 0.0;
                        sv[p] = fma(dm, U[p + 1], sv[p]);
                        cell(p, a, b) = qdn + sv[p];
                        qbd[p] = qdn;
                    }
            }
        }
    }

    // Gam[p][q] = dL/dkappa(x_p, y_q) from Lam (adjoint of the double increment, signature_algs.py:26)
    GPSIG_HD double gamma_at(int p, int q) const {
        if (A.mode == MODE_PT_NODIFF) return cell(0, p, q);
        double g = 0.0;
        if (p > 0 && q > 0) g += cell(0, p - 1, q - 1);
        if (p > 0 && q < R2) g -= cell(0, p - 1, q);
        if (p < R1 && q > 0) g -= cell(0, p, q - 1);
        if (p < R1 && q < R2) g += cell(0, p, q);
        return g;
    }

    // ---- phase C: contract Lam into gradients of the observations ---------------------------------------------
    GPSIG_HD void contract() const {
        const bool yuni = !A.diag;            // Gram launch: the whole wavefront shares y_j
        double* gy = A.diag ? A.gxT : A.gyT;
        const int64_t gys = A.diag ? A.xstride : A.ystride;
        if (A.mode == MODE_INC) {
            // d dM[a][b] / d (increment a of x) = increment b of y, and vice versa; the adjoint of the increment is a difference
            double prev[DP], y0[DP], y1[DP];
#pragma unroll
            for (int f = 0; f < DP; ++f) prev[f] = 0.0;
            for (int a = 0; a <= R1; ++a) {
                double acc[DP];
#pragma unroll
                for (int f = 0; f < DP; ++f) acc[f] = 0.0;
                if (a < R1) {
                    load_y(0, y1);
                    for (int b = 0; b < R2; ++b) {
#pragma unroll
                        for (int f = 0; f < DP; ++f) y0[f] = y1[f];
                        load_y(b + 1, y1);
                        const double lam = cell(0, a, b);
#pragma unroll
                        for (int f = 0; f < DP; ++f) acc[f] = fma(lam, y1[f] - y0[f], acc[f]);
                    }
                }
#pragma unroll
                for (int f = 0; f < DP; ++f) {
                    grad_add(&A.gxT[(int64_t(a) * DP + f) * A.xstride + i], prev[f] - acc[f], false, valid);
                    prev[f] = acc[f];
                }
            }
#pragma unroll
            for (int f = 0; f < DP; ++f) prev[f] = 0.0;
            double x0[DP], x1[DP];
            for (int b = 0; b <= R2; ++b) {
                double acc[DP];
#pragma unroll
                for (int f = 0; f < DP; ++f) acc[f] = 0.0;
                if (b < R2) {
                    load_x(0, x1);
                    for (int a = 0; a < R1; ++a) {
#pragma unroll
                        for (int f = 0; f < DP; ++f) x0[f] = x1[f];
                        load_x(a + 1, x1);
                        const double lam = cell(0, a, b);
#pragma unroll
                        for (int f = 0; f < DP; ++f) acc[f] = fma(lam, x1[f] - x0[f], acc[f]);
                    }
                }
#pragma unroll
                for (int f = 0; f < DP; ++f) {
                    grad_add(&gy[(int64_t(b) * DP + f) * gys + j], prev[f] - acc[f], yuni, valid);
                    prev[f] = acc[f];
                }
            }
            return;
        }
        double gp0 = 0.0;
        double xp[DP], yq[DP];
        for (int p = 0; p < A.L1; ++p) {          // gradient of x_p
            load_x(p, xp);
            const double xs = dot(xp, xp);
            double accy[DP], accx = 0.0;
#pragma unroll
            for (int f = 0; f < DP; ++f) accy[f] = 0.0;
            for (int q = 0; q < A.L2; ++q) {
                load_y(q, yq);
                const double gam = gamma_at(p, q);
                const BaseGrad g = base_eval_grad(A.kind, dot(xp, yq), xs, dot(yq, yq), A.p0, A.p1);
                const double wy = gam * (g.cy - g.cd);
                accx = fma(gam, g.cx + g.cd, accx);
                gp0 = fma(gam, g.dp0, gp0);
#pragma unroll
                for (int f = 0; f < DP; ++f) accy[f] = fma(wy, yq[f], accy[f]);
            }
#pragma unroll
            for (int f = 0; f < DP; ++f) grad_add(&A.gxT[(int64_t(p) * DP + f) * A.xstride + i], fma(accx, xp[f], accy[f]), false, valid);
        }
        for (int q = 0; q < A.L2; ++q) {          // gradient of y_q
            load_y(q, yq);
            const double ys = dot(yq, yq);
            double accx_[DP], accy_ = 0.0;
#pragma unroll
            for (int f = 0; f < DP; ++f) accx_[f] = 0.0;
            for (int p = 0; p < A.L1; ++p) {
                load_x(p, xp);
                const double gam = gamma_at(p, q);
                const BaseGrad g = base_eval_grad(A.kind, dot(xp, yq), dot(xp, xp), ys, A.p0, A.p1);
                const double wx = gam * (g.cy - g.cd);
                accy_ = fma(gam, g.cx2 + g.cd, accy_);
#pragma unroll
                for (int f = 0; f < DP; ++f) accx_[f] = fma(wx, xp[f], accx_[f]);
            }
#pragma unroll
            for (int f = 0; f < DP; ++f) grad_add(&gy[(int64_t(q) * DP + f) * gys + j], fma(accy_, yq[f], accx_[f]), yuni, valid);
        }
        if (A.gbase) grad_add(&A.gbase[0], gp0, true, valid);
    }
};

// ---- tensor vs sequence (signature_algs.py:101-127, kernels.py:313-340) -------------------------------------------
// Level i uses components k0 .. k0+i-1 (k0 = i(i-1)/2).  With m_k[tau] the time increment (difference=True) of
// kz_k(x) = kappa(z_k, x)  [increments=True: kappa(z_k^1, x) - kappa(z_k^0, x), kernels.py:328-330] and u_0 == 1:
//     R_j[tau] = m_{k0+j-1}[tau] * u_{j-1}[tau],   u_j[tau] = sum_{tau' < tau} R_j[tau'],   K_i = sum_tau R_i[tau]
//     w_i == c_i,  w_j[tau] = sum_{tau' > tau} m_{k0+j}[tau'] w_{j+1}[tau'],   dL/dm_{k0+j-1}[tau] = u_{j-1}[tau] * w_j[tau]
// Scratch per pair: slots [0, lt): m_k[tau] then dL/dm_k[tau];  slots [lt, lt + M(M-1)/2): u_j[tau] (j = 1..i-1 of level i).
struct TvsGradArgs {
    const double* z;        // scaled components (lt, T, DP) or (lt, T, 2, DP), features padded to DP
    const double* xT;       // scaled sequences, time-major [(t * DP + f) * xstride + n]
    double* gz;             // same layout as z, accumulated
    double* gxT;            // same layout as xT, accumulated
    int64_t xstride;
    int T, N, L, M, kind, incr, diff;
    int order;              // > 1: higher-order chains (signature_algs.py:129-160), forward_ho / backward_ho
    double p0, p1;
    int t0, nt;             // this launch covers tensors t0 .. t0+nt-1
    const double* G;        // G[m * gm + t * gt + n * gn], m = 0..M
    int64_t gm, gt, gn;
    double* scratch;
    int64_t pairs;
    double* levels;         // optional (M+1) x pairs
    double* gbase;
    int64_t level_t, level_n;   // fused variant: levels[m * pairs + t * level_t + n * level_n]
    GPSIG_HD int64_t pairs_index(int t_, int n_, int m) const { return int64_t(m) * pairs + t_ * level_t + n_ * level_n; }
};

template <int DP>
struct TvsPairGrad {
    const TvsGradArgs& A;
    int t, n;
    int64_t pidx;
    bool valid;
    int R, lt;

    GPSIG_HD TvsPairGrad(const TvsGradArgs& a, int t_, int n_, int64_t pidx_, bool valid_) : A(a), t(t_), n(n_), pidx(pidx_), valid(valid_) {
        R = A.diff ? A.L - 1 : A.L;
        lt = A.M * (A.M + 1) / 2;
    }
    GPSIG_HD double& cell(int s, int tau) const { return A.scratch[(int64_t(s) * R + tau) * A.pairs + pidx]; }
    GPSIG_HD void load_x(int tt, double (&v)[DP]) const {
#pragma unroll
        for (int f = 0; f < DP; ++f) v[f] = A.xT[(int64_t(tt) * DP + f) * A.xstride + n];
    }
    GPSIG_HD const double* zptr(int k, int which) const {
        return A.z + ((int64_t(k) * A.T + t) * (A.incr ? 2 : 1) + which) * DP;
    }
    GPSIG_HD double* gzptr(int k, int which) const {
        return A.gz + ((int64_t(k) * A.T + t) * (A.incr ? 2 : 1) + which) * DP;
    }
    GPSIG_HD double kz(int k, const double (&x)[DP], double xs) const {
        double in1 = 0.0, z1s = 0.0;
        const double* z1 = zptr(k, A.incr ? 1 : 0);
#pragma unroll
        for (int f = 0; f < DP; ++f) { in1 = fma(z1[f], x[f], in1); z1s = fma(z1[f], z1[f], z1s); }
        double v = base_eval<double>(A.kind, in1, z1s, xs, A.p0, A.p1);
        if (A.incr) {
            double in0 = 0.0, z0s = 0.0;
            const double* z0 = zptr(k, 0);
#pragma unroll
            for (int f = 0; f < DP; ++f) { in0 = fma(z0[f], x[f], in0); z0s = fma(z0[f], z0[f], z0s); }
            v -= base_eval<double>(A.kind, in0, z0s, xs, A.p0, A.p1);
        }
        return v;
    }

    GPSIG_HD void forward_m() const {
        // slots [0, lt): m_k[tau]
        double x[DP];
        for (int k = 0; k < lt; ++k) {
            double prev = 0.0;
            for (int tt = 0; tt < A.L; ++tt) {
                load_x(tt, x);
                double xs = 0.0;
#pragma unroll
                for (int f = 0; f < DP; ++f) xs = fma(x[f], x[f], xs);
                const double v = kz(k, x, xs);
                if (A.diff) {
                    if (tt > 0) cell(k, tt - 1) = v - prev;
                    prev = v;
                } else {
                    cell(k, tt) = v;
                }
            }
        }
    }

    GPSIG_HD void forward() const {
        forward_m();
        if (A.levels && valid) A.levels[pidx] = 1.0;
        int k0 = 0, us = lt;
        for (int i = 1; i <= A.M; ++i) {
            // chains of level i: u_1 .. u_{i-1} stored at slots us .. us+i-2
            double u[GRAD_MAX_LEVELS + 1];
#pragma unroll
            for (int jj = 0; jj <= GRAD_MAX_LEVELS; ++jj) u[jj] = 0.0;
            for (int tau = 0; tau < R; ++tau) {
                double carry = 1.0;     // u_0
#pragma unroll
                for (int jj = 1; jj <= GRAD_MAX_LEVELS; ++jj)
                    if (jj <= i) {
                        const double r = cell(k0 + jj - 1, tau) * carry;
                        carry = u[jj];
                        if (jj < i) cell(us + jj - 1, tau) = u[jj];
                        u[jj] += r;
                    }
            }
            double ki = 0.0;
#pragma unroll
            for (int jj = 1; jj <= GRAD_MAX_LEVELS; ++jj)
                if (jj == i) ki = u[jj];
            if (A.levels && valid) A.levels[int64_t(i) * A.pairs + pidx] = ki;
            k0 += i;
            us += i - 1;
        }
    }

    GPSIG_HD void backward() const {
        int k0 = 0, us = lt;
        for (int i = 1; i <= A.M; ++i) {
            const double c = valid ? A.G[i * A.gm + t * A.gt + n * A.gn] : 0.0;
            double w[GRAD_MAX_LEVELS + 2];
#pragma unroll
            for (int jj = 0; jj <= GRAD_MAX_LEVELS + 1; ++jj) w[jj] = 0.0;
            // w[jj] = w_jj[tau] (exclusive suffix), jj = 1..i-1;  w_i == c
            for (int tau = R - 1; tau >= 0; --tau) {
#pragma unroll
                for (int jj = 1; jj <= GRAD_MAX_LEVELS; ++jj)
                    if (jj <= i) {
                        const double wj = jj == i ? c : w[jj];
                        const double m = cell(k0 + jj - 1, tau);
                        const double ujm1 = jj == 1 ? 1.0 : cell(us + jj - 2, tau);
                        cell(k0 + jj - 1, tau) = ujm1 * wj;              // dL/dm
                        if (jj >= 2) w[jj - 1] = fma(m, wj, w[jj - 1]);  // feeds w_{jj-1}[tau-1]; w_jj[tau] was read before any update
                    }
            }
            k0 += i;
            us += i - 1;
        }
    }


    // ---- higher-order chains (signature_kern_tens_vs_seq_higher_order, signature_algs.py:129-160) ---------------------
    // Level i, chain position j = 0 .. i-1 (component k0 + j), D_j = min(j + 1, order) repeat counts l:
    //     R_0[0] = m_{k0};   R_j[0] = m_{k0+j} * P_j,  P_j = excumsum_tau( sum_l R_{j-1}[l] )              (:153)
    //                        R_j[l] = m_{k0+j} * R_{j-1}[l-1] / (l + 1),  l = 1 .. D_j - 1                  (:155)
    //     K_i = sum_tau sum_l R_{i-1}[l]                                                                   (:158)
    // Reverse mode, U_j[l] := dL/dR_j[l]:  U_{i-1}[l] = c_i;
    //     dL/dm_{k0+j} = U_j[0] P_j + sum_{l>=1} U_j[l] R_{j-1}[l-1] / (l + 1)
    //     U_{j-1}[l'] = revexcumsum_tau( m_{k0+j} U_j[0] ) + [l' + 1 < D_j] m_{k0+j} U_j[l'+1] / (l' + 2)
    // Scratch per pair: slots [0, lt): m_k, then dL/dm_k;  per level (reused): R_j[l] at ho_slot(j, l) for j = 0 .. i-2, P_j at
    // ho_pslot(j) for j = 1 .. i-1.  U_j[l] overwrites R_j[l] once position j+1 has used it.
    GPSIG_HD int ho_d(int j) const { return j + 1 < A.order ? j + 1 : A.order; }
    GPSIG_HD int ho_slot(int j, int l) const {
        int s_ = lt;
        for (int jj = 0; jj < j; ++jj) s_ += ho_d(jj);
        return s_ + l;
    }
    GPSIG_HD int ho_pslot(int j) const {           // after the R slots of the deepest level
        int s_ = lt;
        for (int jj = 0; jj + 1 < A.M; ++jj) s_ += ho_d(jj);
        return s_ + j - 1;
    }
    static GPSIG_HD int ho_slots(int M, int order) {
        int s_ = M * (M + 1) / 2;
        for (int jj = 0; jj + 1 < M; ++jj) s_ += (jj + 1 < order ? jj + 1 : order);
        return s_ + (M > 1 ? M - 1 : 0);
    }

    GPSIG_HD void forward_ho() const {
        forward_m();
        if (A.levels && valid) A.levels[pidx] = 1.0;
        int k0 = 0;
        for (int i = 1; i <= A.M; ++i) {
            double ki = 0.0;
            if (i == 1) {
                for (int tau = 0; tau < R; ++tau) ki += cell(k0, tau);
            } else {
                for (int tau = 0; tau < R; ++tau) cell(ho_slot(0, 0), tau) = cell(k0, tau);          // R_0[0]
                for (int j = 1; j < i; ++j) {
                    const int dp = ho_d(j - 1), dc = ho_d(j);
                    double run = 0.0;
                    for (int tau = 0; tau < R; ++tau) {
                        const double m = cell(k0 + j, tau);
                        double sum = 0.0, prev[GRAD_MAX_LEVELS];
                        for (int l = 0; l < dp; ++l) { prev[l] = cell(ho_slot(j - 1, l), tau); sum += prev[l]; }
                        cell(ho_pslot(j), tau) = run;                                                 // P_j[tau]
                        if (j < i - 1) {
                            cell(ho_slot(j, 0), tau) = m * run;
                            for (int l = 1; l < dc; ++l) cell(ho_slot(j, l), tau) = (m * (1.0 / double(l + 1))) * prev[l - 1];
                        } else {                                                                       // last position: only K_i is needed
                            ki += m * run;
                            for (int l = 1; l < dc; ++l) ki += (m * (1.0 / double(l + 1))) * prev[l - 1];
                        }
                        run += sum;
                    }
                }
            }
            if (A.levels && valid) A.levels[int64_t(i) * A.pairs + pidx] = ki;
            // the backward pass of this level runs right away: the level scratch is reused by the next level
            backward_ho_level(i, k0);
            k0 += i;
        }
    }

    GPSIG_HD void backward_ho_level(int i, int k0) const {
        const double c = valid ? A.G[i * A.gm + t * A.gt + n * A.gn] : 0.0;
        if (i == 1) {
            for (int tau = 0; tau < R; ++tau) cell(k0, tau) = c;
            return;
        }
        for (int j = i - 1; j >= 1; --j) {
            const int dp = ho_d(j - 1), dc = ho_d(j);
            double suf = 0.0;                                   // revexcumsum of m * U_j[0]
            for (int tau = R - 1; tau >= 0; --tau) {
                const double m = cell(k0 + j, tau);
                double U[GRAD_MAX_LEVELS];
                for (int l = 0; l < dc; ++l) U[l] = (j == i - 1) ? c : cell(ho_slot(j, l), tau);
                double prev[GRAD_MAX_LEVELS];
                for (int l = 0; l < dp; ++l) prev[l] = cell(ho_slot(j - 1, l), tau);
                double gm_ = U[0] * cell(ho_pslot(j), tau);
                for (int l = 1; l < dc; ++l) gm_ = fma(U[l] * (1.0 / double(l + 1)), prev[l - 1], gm_);
                cell(k0 + j, tau) = gm_;                                                               // dL/dm_{k0+j}
                for (int l = 0; l < dp; ++l) {
                    double u = suf;
                    if (l + 1 < dc) u = fma(m * (1.0 / double(l + 2)), U[l + 1], u);
                    cell(ho_slot(j - 1, l), tau) = u;                                                  // U_{j-1}[l] over R_{j-1}[l]
                }
                suf = fma(m, U[0], suf);
            }
        }
        for (int tau = 0; tau < R; ++tau) cell(k0, tau) = cell(ho_slot(0, 0), tau);                   // dL/dm_{k0} = U_0[0]
    }

    // gk(k, tt) = dL/d kz_k(x_tt)
    GPSIG_HD double gk_at(int k, int tt) const {
        if (!A.diff) return cell(k, tt);
        double g = 0.0;
        if (tt > 0) g += cell(k, tt - 1);
        if (tt < R) g -= cell(k, tt);
        return g;
    }

    GPSIG_HD void contract() const {
        double x[DP];
        double gp0 = 0.0;
        // gradient of the observations: time outer, components inner
        for (int tt = 0; tt < A.L; ++tt) {
            load_x(tt, x);
            double xs = 0.0;
#pragma unroll
            for (int f = 0; f < DP; ++f) xs = fma(x[f], x[f], xs);
            double accz[DP], accx = 0.0;
#pragma unroll
            for (int f = 0; f < DP; ++f) accz[f] = 0.0;
            for (int k = 0; k < lt; ++k) {
                const double gk = gk_at(k, tt);
                for (int which = (A.incr ? 0 : 0); which < (A.incr ? 2 : 1); ++which) {
                    const double sgn = (A.incr && which == 0) ? -1.0 : 1.0;
                    const double* z = zptr(k, which);
                    double in = 0.0, zs = 0.0;
#pragma unroll
                    for (int f = 0; f < DP; ++f) { in = fma(z[f], x[f], in); zs = fma(z[f], z[f], zs); }
                    const BaseGrad g = base_eval_grad(A.kind, in, zs, xs, A.p0, A.p1);   // first argument z, second x
                    const double gg = sgn * gk;
                    const double wz = gg * (g.cy - g.cd);
                    accx = fma(gg, g.cx2 + g.cd, accx);
                    gp0 = fma(gg, g.dp0, gp0);
#pragma unroll
                    for (int f = 0; f < DP; ++f) accz[f] = fma(wz, z[f], accz[f]);
                }
            }
#pragma unroll
            for (int f = 0; f < DP; ++f) grad_add(&A.gxT[(int64_t(tt) * DP + f) * A.xstride + n], fma(accx, x[f], accz[f]), false, valid);
        }
        // gradient of the tensor components: components outer, time inner
        for (int k = 0; k < lt; ++k) {
            for (int which = 0; which < (A.incr ? 2 : 1); ++which) {
                const double sgn = (A.incr && which == 0) ? -1.0 : 1.0;
                const double* z = zptr(k, which);
                double zs = 0.0;
#pragma unroll
                for (int f = 0; f < DP; ++f) zs = fma(z[f], z[f], zs);
                double accx_[DP], accz_ = 0.0;
#pragma unroll
                for (int f = 0; f < DP; ++f) accx_[f] = 0.0;
                for (int tt = 0; tt < A.L; ++tt) {
                    load_x(tt, x);
                    double in = 0.0, xs = 0.0;
#pragma unroll
                    for (int f = 0; f < DP; ++f) { in = fma(z[f], x[f], in); xs = fma(x[f], x[f], xs); }
                    const BaseGrad g = base_eval_grad(A.kind, in, zs, xs, A.p0, A.p1);
                    const double gg = sgn * gk_at(k, tt);
                    const double wx = gg * (g.cy - g.cd);
                    accz_ = fma(gg, g.cx + g.cd, accz_);
#pragma unroll
                    for (int f = 0; f < DP; ++f) accx_[f] = fma(wx, x[f], accx_[f]);
                }
                double* gzp = gzptr(k, which);
#pragma unroll
                for (int f = 0; f < DP; ++f) grad_add(&gzp[f], fma(accz_, z[f], accx_[f]), true, valid);
            }
        }
        if (A.gbase) grad_add(&A.gbase[0], gp0, true, valid);
    }
};

// The same gradient without any scratch memory: level by level, the chain prefixes u_j are rebuilt on the way back by
// undoing the forward recursion (u_j[tau] = u_j[tau+1] - m[tau] * u_{j-1}[tau], lowest chain first), the base kernel is
// evaluated once per (component, time point) and pass, and the contraction with its derivatives happens in the same
// backward sweep.  MMAX bounds the chain length (registers), E = 2 for increments.
//
// tvs_level_grad handles ONE level of ONE (tensor, sequence) pair.  Where the data comes from and where the gradient of
// the observations goes is the caller's business (IO): the one-pair-per-thread kernel reads a time-major array and adds
// with atomics, the tensor-lane kernel reads both operands from LDS and reduces over the wavefront first.
//   IO::z(k, e, f): feature f of component k, point e;  IO::zsq(k, e): its squared norm
//   IO::load_x(tt, v) -> |x_tt|^2     IO::emit_gx(tt, gx)
//   IO::sign() / IO::combine(k): +1 / identity, except where the two points of an incremental tensor sit in two adjacent
//   lanes (E == 1 per lane): then sign() is -1 for the first point and combine() adds the partner lane's value.
template <int E>
struct TvsEv {                // kz_k(x) = sum_e sign_e kappa(z_k^e, x) and what its derivatives need
    double k;
    double wz[E];             // sign * (cy - cd): multiplies z in d/dx and x in d/dz
    double vx[E];             // sign * (cx2 + cd): multiplies x in d/dx
    double vz[E];             // sign * (cx + cd): multiplies z in d/dz
    double dp0;               // d kz / d base_params[0]
};

// KIND >= 0 fixes the base kernel at compile time (the derivative coefficients that are identically zero, or equal to
// the kernel value, then cost no registers); KIND < 0 takes it from the argument.
template <int DP, int E, int KIND, class IO>
GPSIG_HD TvsEv<E> tvs_eval(const IO& io, int k, const double (&x)[DP], double xs, bool with_grad, int kind_rt, double p0, double p1) {
    const int kind = KIND >= 0 ? KIND : kind_rt;
    TvsEv<E> r;
    r.k = 0.0;
    r.dp0 = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const double sgn = E == 2 ? (e == 0 ? -1.0 : 1.0) : io.sign();
        double in = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f) in = fma(io.z(k, e, f), x[f], in);
        const double zs = io.zsq(k, e);
        if (with_grad) {
            const BaseGrad g = base_eval_grad(kind, in, zs, xs, p0, p1);
            r.k += sgn * g.k;
            r.wz[e] = sgn * (g.cy - g.cd);
            r.vx[e] = sgn * (g.cx2 + g.cd);
            r.vz[e] = sgn * (g.cx + g.cd);
            r.dp0 += sgn * g.dp0;
        } else {
            r.k += sgn * base_eval<double>(kind, in, zs, xs, p0, p1);
            r.wz[e] = r.vx[e] = r.vz[e] = 0.0;
        }
    }
    r.k = io.combine(r.k);
    return r;
}

// KIND as in tvs_eval: the linear kernel has d kz/dx = wz z, d kz/dz = wz x; the RBF kernel d kz/dx = wz (z - x) = -d kz/dz.
template <int DP, int MMAX, int E, int IC = 0, int KIND = -1, class IO>
GPSIG_HD void tvs_contract(IO& io, int i_rt, int k0, int tt, const double (&x)[DP], const double (&gk)[MMAX], const TvsEv<E> (&ev)[MMAX],
                           double (&gzacc)[MMAX][E][DP], double& gp0) {
    const int i = IC > 0 ? IC : i_rt;
    double gx[DP];
#pragma unroll
    for (int f = 0; f < DP; ++f) gx[f] = 0.0;
#pragma unroll
    for (int j = 0; j < MMAX; ++j)
        if (j < i) {
            gp0 = fma(gk[j], ev[j].dp0, gp0);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double a = gk[j] * ev[j].wz[e];
                if constexpr (KIND == BASE_LINEAR) {
#pragma unroll
                    for (int f = 0; f < DP; ++f) {
                        gx[f] = fma(a, io.z(k0 + j, e, f), gx[f]);
                        gzacc[j][e][f] = fma(a, x[f], gzacc[j][e][f]);
                    }
                } else if constexpr (KIND == BASE_RBF) {
                    const double na = -a;
#pragma unroll
                    for (int f = 0; f < DP; ++f) {
                        const double dzx = io.z(k0 + j, e, f) - x[f];
                        gx[f] = fma(a, dzx, gx[f]);
                        gzacc[j][e][f] = fma(na, dzx, gzacc[j][e][f]);
                    }
                } else {
                    const double bx = gk[j] * ev[j].vx[e], bz = gk[j] * ev[j].vz[e];
#pragma unroll
                    for (int f = 0; f < DP; ++f) {
                        const double zf = io.z(k0 + j, e, f);
                        gx[f] = fma(a, zf, fma(bx, x[f], gx[f]));
                        gzacc[j][e][f] = fma(a, x[f], fma(bz, zf, gzacc[j][e][f]));
                    }
                }
            }
        }
    io.emit_gx(tt, gx);
}

// i: level (chain length), k0: its first component, R: number of increments (or of points when diff is false),
// c: upstream gradient of this level.  Returns the level's value (the forward result, free of charge).
// IC > 0 fixes the chain length at compile time (the loops over the chains then carry no predication).
template <int DP, int MMAX, int E, int KIND = -1, int IC = 0, class IO>
GPSIG_HD double tvs_level_grad(IO& io, int i_rt, int k0, int R, bool diff, int kind, double p0, double p1, double c,
                               double (&gzacc)[MMAX][E][DP], double& gp0) {
    const int i = IC > 0 ? IC : i_rt;
    // ---- forward: chains u_1 .. u_i (exclusive prefixes; after the sweep u_j = sum_tau R_j[tau])
    double u[MMAX + 1];
#pragma unroll
    for (int j = 0; j <= MMAX; ++j) u[j] = 0.0;
    double x[DP], kprev[MMAX];
    if (diff) {
        const double xs = io.load_x(0, x);
#pragma unroll
        for (int j = 0; j < MMAX; ++j) kprev[j] = j < i ? tvs_eval<DP, E, KIND>(io, k0 + j, x, xs, false, kind, p0, p1).k : 0.0;
    }
    for (int tau = 0; tau < R; ++tau) {
            const double xs = io.load_x(diff ? tau + 1 : tau, x);
        double carry = 1.0;
#pragma unroll
        for (int j = 0; j < MMAX; ++j)
            if (j < i) {
                const double kv = tvs_eval<DP, E, KIND>(io, k0 + j, x, xs, false, kind, p0, p1).k;
                const double m = diff ? kv - kprev[j] : kv;
                kprev[j] = kv;
                const double r = m * carry;
                carry = u[j + 1];
                u[j + 1] += r;
            }
    }
    double ki = 0.0;
#pragma unroll
    for (int j = 1; j <= MMAX; ++j)
        if (j == i) ki = u[j];
    // ---- backward: undo the chains, build w, contract
    double w[MMAX + 1], gprev[MMAX];
    TvsEv<E> evn[MMAX];      // evaluations at the later time point (tau + 1)
#pragma unroll
    for (int j = 0; j <= MMAX; ++j) w[j] = 0.0;
#pragma unroll
    for (int j = 0; j < MMAX; ++j) gprev[j] = 0.0;
    double xn[DP];           // x at the later time point
    if (diff) {
        const double xs = io.load_x(R, xn);
#pragma unroll
        for (int j = 0; j < MMAX; ++j)
            if (j < i) evn[j] = tvs_eval<DP, E, KIND>(io, k0 + j, xn, xs, true, kind, p0, p1);
    }
    for (int tau = R - 1; tau >= 0; --tau) {
            const double xs = io.load_x(tau, x);
        TvsEv<E> evc[MMAX];
        double m[MMAX], gm[MMAX], ulow[MMAX];
#pragma unroll
        for (int j = 0; j < MMAX; ++j)
            if (j < i) {
                evc[j] = tvs_eval<DP, E, KIND>(io, k0 + j, x, xs, true, kind, p0, p1);
                m[j] = diff ? evn[j].k - evc[j].k : evc[j].k;
            }
        // undo: u_{j+1}[tau] = u_{j+1}[tau+1] - m[j] * u_j[tau]   (u_0 == 1), lowest chain first
        double below = 1.0;
#pragma unroll
        for (int j = 0; j < MMAX; ++j)
            if (j < i) {
                ulow[j] = below;                 // u_j[tau]: what R_{j+1}[tau] was multiplied with
                u[j + 1] = fma(-m[j], below, u[j + 1]);
                below = u[j + 1];
            }
#pragma unroll
        for (int j = 0; j < MMAX; ++j)
            if (j < i) gm[j] = ulow[j] * (j == i - 1 ? c : w[j + 1]);           // dL/dm_{k0+j}[tau] = u_j[tau] * w_{j+1}[tau]
#pragma unroll
        for (int j = 1; j < MMAX; ++j)
            if (j < i) w[j] = fma(m[j], (j == i - 1 ? c : w[j + 1]), w[j]);     // w_j[tau-1] += m_{k0+j}[tau] * w_{j+1}[tau]
        if (diff) {
            double gk[MMAX];
#pragma unroll
            for (int j = 0; j < MMAX; ++j) gk[j] = j < i ? gm[j] - gprev[j] : 0.0;   // dL/d kz(x_{tau+1})
            tvs_contract<DP, MMAX, E, IC, KIND>(io, i, k0, tau + 1, xn, gk, evn, gzacc, gp0);
#pragma unroll
            for (int j = 0; j < MMAX; ++j)
                if (j < i) { gprev[j] = gm[j]; evn[j] = evc[j]; }
#pragma unroll
            for (int f = 0; f < DP; ++f) xn[f] = x[f];
        } else {
            tvs_contract<DP, MMAX, E, IC, KIND>(io, i, k0, tau, x, gm, evc, gzacc, gp0);
        }
    }
    if (diff) {          // time point 0
        double gk[MMAX];
#pragma unroll
        for (int j = 0; j < MMAX; ++j) gk[j] = j < i ? -gprev[j] : 0.0;
        tvs_contract<DP, MMAX, E, IC, KIND>(io, i, k0, 0, xn, gk, evn, gzacc, gp0);
    }
    return ki;
}

// one (tensor, sequence) pair per thread: operands from global memory, atomics for the observations
template <int DP, int MMAX, int E>
struct TvsPairGradFused {
    const TvsGradArgs& A;
    int t, n;
    bool valid;

    GPSIG_HD TvsPairGradFused(const TvsGradArgs& a, int t_, int n_, bool valid_) : A(a), t(t_), n(n_), valid(valid_) {}
    GPSIG_HD double z(int k, int e, int f) const { return A.z[((int64_t(k) * A.T + t) * E + e) * DP + f]; }
    GPSIG_HD double sign() const { return 1.0; }
    GPSIG_HD double combine(double k) const { return k; }
    GPSIG_HD double zsq(int k, int e) const {
        double s = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f) s = fma(z(k, e, f), z(k, e, f), s);
        return s;
    }
    GPSIG_HD double load_x(int tt, double (&v)[DP]) const {
        double s = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f) {
            v[f] = A.xT[(int64_t(tt) * DP + f) * A.xstride + n];
            s = fma(v[f], v[f], s);
        }
        return s;
    }
    GPSIG_HD void emit_gx(int tt, const double (&gx)[DP]) const {
#pragma unroll
        for (int f = 0; f < DP; ++f) grad_add(&A.gxT[(int64_t(tt) * DP + f) * A.xstride + n], gx[f], false, valid);
    }
    GPSIG_HD void run() {
        const int R = A.diff ? A.L - 1 : A.L;
        int k0 = 0;
        double gp0 = 0.0;
        if (A.levels && valid) A.levels[A.pairs_index(t, n, 0)] = 1.0;
        for (int i = 1; i <= A.M; ++i) {
            double gzacc[MMAX][E][DP];
#pragma unroll
            for (int j = 0; j < MMAX; ++j)
#pragma unroll
                for (int e = 0; e < E; ++e)
#pragma unroll
                    for (int f = 0; f < DP; ++f) gzacc[j][e][f] = 0.0;
            const double c = valid ? A.G[i * A.gm + t * A.gt + n * A.gn] : 0.0;
            const double ki = tvs_level_grad<DP, MMAX, E>(*this, i, k0, R, A.diff != 0, A.kind, A.p0, A.p1, c, gzacc, gp0);
            if (A.levels && valid) A.levels[A.pairs_index(t, n, i)] = ki;
#pragma unroll
            for (int j = 0; j < MMAX; ++j)
                if (j < i) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        double* gzp = A.gz + ((int64_t(k0 + j) * A.T + t) * E + e) * DP;
#pragma unroll
                        for (int f = 0; f < DP; ++f) grad_add(&gzp[f], gzacc[j][e][f], true, valid);
                    }
                }
            k0 += i;
        }
        if (A.gbase) grad_add(&A.gbase[0], gp0, true, valid);
    }
};

// ---- tensor vs tensor (signature_algs.py:76-99, kernels.py:263-283) -------------------------------------------------
// Level i = prod_{j<i} Mz_{k0+j}[t][t'],  Mz_k = kappa(z_k[t], z_k[t'])  or, with increments (:275-277),
// kappa(z1,z1') + kappa(z0,z0') - kappa(z1,z0') - kappa(z0,z1').  One (t, t') entry per thread.
struct TensGradArgs {
    const double* z;     // scaled components (lt, T, DP) or (lt, T, 2, DP)
    double* gz;          // accumulated
    int T, M, kind, incr;
    double p0, p1;
    const double* G;     // G[m * gm + t * gt + t2 * gn], m = 0..M
    int64_t gm, gt, gn;
    double* gbase;
    double* part;        // row-owned kernel: per-slice partial sums (nslices, rows, DP) instead of atomics on gz, or null
    int64_t part_stride; // rows * DP
};

template <int DP>
struct TensPairGrad {
    const TensGradArgs& A;
    int t, t2;
    bool valid;

    GPSIG_HD TensPairGrad(const TensGradArgs& a, int t_, int t2_, bool valid_) : A(a), t(t_), t2(t2_), valid(valid_) {}
    GPSIG_HD const double* zptr(int k, int tt, int which) const { return A.z + ((int64_t(k) * A.T + tt) * (A.incr ? 2 : 1) + which) * DP; }
    GPSIG_HD double* gzptr(int k, int tt, int which) const { return A.gz + ((int64_t(k) * A.T + tt) * (A.incr ? 2 : 1) + which) * DP; }
    GPSIG_HD BaseGrad kg(const double* a, const double* b) const {
        double in = 0.0, as = 0.0, bs = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f) { in = fma(a[f], b[f], in); as = fma(a[f], a[f], as); bs = fma(b[f], b[f], bs); }
        return base_eval_grad(A.kind, in, as, bs, A.p0, A.p1);
    }
    GPSIG_HD double mz(int k) const {
        if (!A.incr) return kg(zptr(k, t, 0), zptr(k, t2, 0)).k;
        return kg(zptr(k, t, 1), zptr(k, t2, 1)).k + kg(zptr(k, t, 0), zptr(k, t2, 0)).k - kg(zptr(k, t, 1), zptr(k, t2, 0)).k -
               kg(zptr(k, t, 0), zptr(k, t2, 1)).k;
    }
    // adds gm * d kappa(z_k[t][wa], z_k[t2][wb]) to both arguments
    GPSIG_HD void scatter(int k, int wa, int wb, double gm, double& gp0) const {
        const double* a = zptr(k, t, wa);
        const double* b = zptr(k, t2, wb);
        const BaseGrad g = kg(a, b);
        gp0 = fma(gm, g.dp0, gp0);
        double* ga = gzptr(k, t, wa);
        double* gb = gzptr(k, t2, wb);
#pragma unroll
        for (int f = 0; f < DP; ++f) {
            const double d = a[f] - b[f];
            grad_add(&ga[f], gm * (g.cy * b[f] + g.cx * a[f] + g.cd * d), true, valid);     // the wavefront shares t
            grad_add(&gb[f], gm * (g.cy * a[f] + g.cx2 * b[f] - g.cd * d), false, valid);
        }
    }
    GPSIG_HD void run() const {
        int k0 = 0;
        double gp0 = 0.0;
        for (int i = 1; i <= A.M; ++i) {
            const double c = valid ? A.G[i * A.gm + t * A.gt + t2 * A.gn] : 0.0;
            double m[GRAD_MAX_LEVELS];
#pragma unroll
            for (int jj = 0; jj < GRAD_MAX_LEVELS; ++jj) m[jj] = jj < i ? mz(k0 + jj) : 1.0;
#pragma unroll
            for (int jj = 0; jj < GRAD_MAX_LEVELS; ++jj)
                if (jj < i) {
                    double others = c;
#pragma unroll
                    for (int q = 0; q < GRAD_MAX_LEVELS; ++q)
                        if (q != jj) others *= m[q];
                    if (!A.incr) {
                        scatter(k0 + jj, 0, 0, others, gp0);
                    } else {
                        scatter(k0 + jj, 1, 1, others, gp0);
                        scatter(k0 + jj, 0, 0, others, gp0);
                        scatter(k0 + jj, 1, 0, -others, gp0);
                        scatter(k0 + jj, 0, 1, -others, gp0);
                    }
                }
            k0 += i;
        }
        if (A.gbase) grad_add(&A.gbase[0], gp0, true, valid);
    }
};

// Row-owned variant: thread = inducing tensor t (and a slice of the partners t2).  Kzz's level products are symmetric under
// t <-> t2, so  d/dz_t sum G[t][t2] F(t, t2) = sum_t2 (G[t][t2] + G[t2][t]) d_1 F(t, t2):  only derivatives with respect to the
// FIRST argument are needed and they accumulate in registers, one component at a time (the other components' kernel values
// are re-evaluated per component -- i times more kernel evaluations on a T x T problem, in exchange for E * DP accumulators
// instead of i * E * DP), so each thread ends with lt * E * DP atomics instead of one wavefront reduction per
// (component, point, feature, partner).
template <int DP, int E>
struct TensRowGrad {
    const TensGradArgs& A;
    int t;
    bool valid;

    GPSIG_HD TensRowGrad(const TensGradArgs& a, int t_, bool valid_) : A(a), t(t_), valid(valid_) {}
    GPSIG_HD const double* zptr(int k, int tt, int e) const { return A.z + ((int64_t(k) * A.T + tt) * E + e) * DP; }
    // Mz_k(t, t2): kappa for E == 1, the double increment of kernels.py:277 for E == 2
    GPSIG_HD double mz(int k, int t2) const {
        double mv = 0.0;
#pragma unroll
        for (int a = 0; a < E; ++a)
#pragma unroll
            for (int b = 0; b < E; ++b) {
                const double* za = zptr(k, t, a);
                const double* zb = zptr(k, t2, b);
                double in = 0.0, as = 0.0, bs = 0.0;
#pragma unroll
                for (int f = 0; f < DP; ++f) { in = fma(za[f], zb[f], in); as = fma(za[f], za[f], as); bs = fma(zb[f], zb[f], bs); }
                mv += (a == b ? 1.0 : -1.0) * base_eval<double>(A.kind, in, as, bs, A.p0, A.p1);
            }
        return mv;
    }
    // slices: partners t2 = slice, slice + nslices, ...;  comp >= 0: only component comp (the launch spreads the components
    // over workgroups: the chain of dependent loads per wavefront is what bounds this kernel at a few hundred tensors)
    GPSIG_HD void run(int slice, int nslices, int comp = -1) const {
        int k0 = 0;
        double gp0 = 0.0;
        for (int i = 1; i <= A.M; ++i) {
            for (int j = 0; j < i; ++j) {
                if (comp >= 0 && k0 + j != comp) continue;
                double za[E][DP], zas[E], acc[E][DP];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    double s = 0.0;
#pragma unroll
                    for (int f = 0; f < DP; ++f) {
                        za[e][f] = zptr(k0 + j, t, e)[f];
                        acc[e][f] = 0.0;
                        s = fma(za[e][f], za[e][f], s);
                    }
                    zas[e] = s;
                }
                for (int t2 = slice; t2 < A.T; t2 += nslices) {
                    const double g12 = valid ? A.G[i * A.gm + t * A.gt + t2 * A.gn] : 0.0;
                    const double c = g12 + (valid ? A.G[i * A.gm + t2 * A.gt + t * A.gn] : 0.0);
                    double others = 1.0;
                    for (int q = 0; q < i; ++q)
                        if (q != j) others *= mz(k0 + q, t2);
#pragma unroll
                    for (int a = 0; a < E; ++a)
#pragma unroll
                        for (int b = 0; b < E; ++b) {
                            const double* zb = zptr(k0 + j, t2, b);
                            double in = 0.0, bs = 0.0;
#pragma unroll
                            for (int f = 0; f < DP; ++f) { in = fma(za[a][f], zb[f], in); bs = fma(zb[f], zb[f], bs); }
                            const BaseGrad g = base_eval_grad(A.kind, in, zas[a], bs, A.p0, A.p1);
                            const double sg = (a == b ? 1.0 : -1.0);
                            gp0 = fma(g12 * others * sg, g.dp0, gp0);
                            const double w = c * others * sg;
                            const double wy = w * (g.cy - g.cd), wx = w * (g.cx + g.cd);     // d kappa/dx = (cy - cd) y + (cx + cd) x
#pragma unroll
                            for (int f = 0; f < DP; ++f) acc[a][f] = fma(wy, zb[f], fma(wx, za[a][f], acc[a][f]));
                        }
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int64_t at = ((int64_t(k0 + j) * A.T + t) * E + e) * DP;
                    if (A.part) {
                        // same-address atomics from hundreds of workgroups queue up at the memory side of the chip (about a
                        // microsecond each): partial sums per slice, added up by tens_row_reduce_kernel
                        if (valid) {
#pragma unroll
                            for (int f = 0; f < DP; ++f) A.part[slice * A.part_stride + at + f] = acc[e][f];
                        }
                    } else {
#pragma unroll
                        for (int f = 0; f < DP; ++f) grad_add(&A.gz[at + f], acc[e][f], false, valid);
                    }
                }
            }
            k0 += i;
        }
        if (A.gbase) grad_add(&A.gbase[0], gp0, true, valid);
    }
};

}  // namespace gpsig
