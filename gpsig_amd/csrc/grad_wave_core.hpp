// grad_wave_core.hpp -- per-lane arithmetic of the wavefront-parallel sequence-pair gradient (the fast path of
// gpsig_seq_gram_levels_grad / gpsig_seq_diag_levels_grad; formulas in the header of grad_core.hpp).
//
// Lane mapping as in the evaluation kernel (seq_core.hpp): a pair occupies G consecutive lanes, lane `lam` owns the C
// lattice columns C*lam .. C*lam+C-1.  Two sweeps per pair:
//   forward  -- lane lam works on lattice row t - lam at step t (row prefixes handed to the RIGHT neighbour one step
//               later), and writes its Q_m[a][b] (m < M) for every row to an HBM scratch array indexed by the step, so
//               that a wavefront's stores are contiguous;
//   backward -- rows descending with the opposite skew (lane lam works on row R1-1 - (u - (G-1-lam)) at step u; row
//               suffixes of dM * U_{m+1} handed to the LEFT neighbour one step later).  The forward value a cell needs,
//               Q_{p-1}[a-1][b-1], was written at forward step a-1+lam -- the same step index for every lane -- so the
//               loads are contiguous too.  Out comes Lam[a][b] = dL/ddM[a][b].
// Nothing here touches memory spaces or lanes: neighbour values, scratch loads and x rows are arguments.  Shared verbatim
// by the gfx950 kernel (grad_wave_kernel.hpp) and the host-side lock-step harness under tests/.
#pragma once

#include "grad_core.hpp"

namespace gpsig {

// how one lane produces dM for its C columns at a lattice row, from point rows of x and its own points of y
template <int C, int DP, int MODE>
struct WaveDm {
    // MODE_INC: dy[c] = y_{b+1} - y_b.  Point modes: yp[c] = y_{b_c} (C+1 of them for MODE_PT_DIFF) and squared norms.
    double y[C + 1][DP];
    double ys[C + 1];
    double rd[C];            // MODE_PT_DIFF: kappa(x_r, y_{b+1}) - kappa(x_r, y_b) of the row kept from the previous step
    double xk[DP];           // x point row kept from the previous step (MODE_INC)
    int nvalid;              // columns of this lane inside the lattice (0..C)

    // ypts: C+1 consecutive point rows of y starting at the lane's first column (rows beyond the sequence: any finite value)
    GPSIG_HD void set_y(const double (&ypts)[C + 1][DP], int nvalid_) {
        nvalid = nvalid_;
#pragma unroll
        for (int c = 0; c <= C; ++c) {
            double s = 0.0;
#pragma unroll
            for (int f = 0; f < DP; ++f) {
                if (MODE == MODE_INC) {
                    if (c < C) y[c][f] = ypts[c + 1][f] - ypts[c][f];
                } else {
                    y[c][f] = ypts[c][f];
                }
                s = fma(ypts[c][f], ypts[c][f], s);
            }
            ys[c] = s;
        }
    }
    GPSIG_HD void row_diff(const double (&x)[DP], int kind, double p0, double p1, double (&out)[C]) const {
        double xs = 0.0, k[C + 1];
#pragma unroll
        for (int f = 0; f < DP; ++f) xs = fma(x[f], x[f], xs);
#pragma unroll
        for (int c = 0; c <= C; ++c) {
            double in = 0.0;
#pragma unroll
            for (int f = 0; f < DP; ++f) in = fma(x[f], y[c][f], in);
            k[c] = base_eval<double>(kind, in, xs, ys[c], p0, p1);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = k[c + 1] - k[c];
    }
    // Before a sweep: the point row the first lattice row of the sweep pairs with (forward: x_0; backward: x_{R1}).
    GPSIG_HD void prime(const double (&x)[DP], int kind, double p0, double p1) {
        if (MODE == MODE_INC) {
#pragma unroll
            for (int f = 0; f < DP; ++f) xk[f] = x[f];
        } else if (MODE == MODE_PT_DIFF) {
            row_diff(x, kind, p0, p1, rd);
        }
    }
    // One lattice row.  xnew: forward sweep x_{a+1}, backward sweep x_a (MODE_PT_NODIFF: x_a in both).
    GPSIG_HD void row(const double (&xnew)[DP], bool forward, int kind, double p0, double p1, double (&dm)[C]) {
        if (MODE == MODE_INC) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int f = 0; f < DP; ++f) acc = fma(forward ? xnew[f] - xk[f] : xk[f] - xnew[f], y[c][f], acc);
                dm[c] = acc;
            }
#pragma unroll
            for (int f = 0; f < DP; ++f) xk[f] = xnew[f];
        } else if (MODE == MODE_PT_DIFF) {
            double nd[C];
            row_diff(xnew, kind, p0, p1, nd);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                dm[c] = forward ? nd[c] - rd[c] : rd[c] - nd[c];
                rd[c] = nd[c];
            }
        } else {
            double xs = 0.0;
#pragma unroll
            for (int f = 0; f < DP; ++f) xs = fma(xnew[f], xnew[f], xs);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                double in = 0.0;
#pragma unroll
                for (int f = 0; f < DP; ++f) in = fma(xnew[f], y[c][f], in);
                dm[c] = base_eval<double>(kind, in, xs, ys[c], p0, p1);
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (c >= nvalid) dm[c] = 0.0;
    }
};

// Forward recursion state of one lane.  LQ = number of levels whose Q is kept (levels 1..M-1, M-1 <= LQ):
// q[m-1][c] = Q_m[a][b_c], qg[m-1] = Q_m[a][b_0 - 1] (ghost column).  sout[m], m = 1..M: end-of-chunk row prefix of R_m.
template <int C, int LQ>
struct WaveFwd {
    double q[LQ][C], qg[LQ], sout[LQ + 2];

    GPSIG_HD void reset() {
#pragma unroll
        for (int m = 0; m < LQ; ++m) {
            qg[m] = 0.0;
#pragma unroll
            for (int c = 0; c < C; ++c) q[m][c] = 0.0;
        }
#pragma unroll
        for (int m = 0; m < LQ + 2; ++m) sout[m] = 0.0;
    }
    // cin[m] (m = 1..M): the left neighbour's sout[m] of ITS previous step; zeros for the first lane of a pair.
    GPSIG_HD void step(const double (&dm)[C], const double (&cin)[LQ + 2], int M) {
#pragma unroll
        for (int m = LQ + 1; m >= 2; --m)
            if (m <= M) {
                constexpr int dummy = 0;
                (void)dummy;
                const int lo = m - 2;                          // level m-1
                const int me = m - 1 < LQ ? m - 1 : LQ - 1;     // level m (only touched when m < M, i.e. m <= LQ)
                double s = cin[m];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    s = fma(dm[c], c == 0 ? qg[lo] : q[lo][c > 0 ? c - 1 : 0], s);      // dM * Q_{m-1}[a-1][b_c - 1]
                    if (m < M) q[me][c] += s;
                }
                sout[m] = s;
                if (m < M) qg[me] += cin[m];
            }
        double s = cin[1];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            s += dm[c];
            if (1 < M) q[0][c] += s;
        }
        sout[1] = s;
        if (1 < M) qg[0] += cin[1];
    }
};

// Backward recursion state: qb[p-1][c] = Qb_p[a+1][b_c] (suffix sums of dM * U_{p+1}), qbg[p-1] = Qb_p[a+1][b_{C-1} + 1]; p = 1..M-1.
template <int C, int LQ>
struct WaveBwd {
    double qb[LQ][C], qbg[LQ], svout[LQ];

    GPSIG_HD void reset() {
#pragma unroll
        for (int p = 0; p < LQ; ++p) {
            qbg[p] = svout[p] = 0.0;
#pragma unroll
            for (int c = 0; c < C; ++c) qb[p][c] = 0.0;
        }
    }
    // clev[p] (p = 1..M): upstream gradient of level p for this pair.  qfd[m-1][c] = Q_m[a-1][b_c - 1], m = 1..M-1 (zeros at the
    // lattice border).  sin[p-1]: the right neighbour's svout[p-1] of ITS previous step; zeros for the last lane of a pair.
    GPSIG_HD void step(const double (&dm)[C], const double (&clev)[LQ + 2], const double (&qfd)[LQ][C], const double (&sin)[LQ], int M,
                       double (&lam)[C]) {
        double U[LQ + 2][C];            // U[p][c], p = 1..M  (U_M == c_M)
#pragma unroll
        for (int p = 1; p <= LQ + 1; ++p)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int pi = p - 1 < LQ ? p - 1 : LQ - 1;
                if (p < M) U[p][c] = clev[p] + (c < C - 1 ? qb[pi][c < C - 1 ? c + 1 : c] : qbg[pi]);
                else U[p][c] = p == M ? clev[p] : 0.0;
            }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            double l = U[1][c];
#pragma unroll
            for (int p = 2; p <= LQ + 1; ++p)
                if (p <= M) l = fma(qfd[p - 2][c], U[p][c], l);
            lam[c] = l;
        }
#pragma unroll
        for (int p = 1; p <= LQ; ++p)
            if (p < M) {
                double sv = sin[p - 1];
#pragma unroll
                for (int c = C - 1; c >= 0; --c) {
                    sv = fma(dm[c], U[p + 1][c], sv);
                    qb[p - 1][c] += sv;
                }
                svout[p - 1] = sv;
                qbg[p - 1] += sin[p - 1];
            }
    }
};

// geometry of one launch of the wave kernel
struct WaveGradArgs {
    const double* X; const double* Y;      // scaled observations, user layout (N, L, d) row-major
    int N1, N2, L1, L2, d;
    int M, kind, mode;
    double p0, p1;
    int diag;                              // pairs (i, i)
    int64_t pair0, npairs;                 // this launch covers pairs pair0 .. pair0+npairs-1 of the enumeration p = i * N2 + j (diag: p = i)
    const double* G; int64_t gm, gi, gj;   // upstream gradient of the levels
    double* scratch;                       // per group slot: (M-1) * TF * G * C doubles
    double* lam;                           // out: Lam, (npairs, R1, R2) row-major
    int ngroups;                           // groups in flight = scratch slots
};

// =====================================================================================================================
// Scratch-free formulation.  The backward sweep does not read the forward Q's back from memory, it UNDOES the forward
// recursion row by row: Q_m[a-1][b] = Q_m[a][b] - s_m[a][b], with the row prefix s_m[a][b] written as
// (row total) - (row suffix).  The row totals rowtot_m[a] = sum_b R_m[a][b] are the only thing the forward sweep has to
// leave behind (M-1 doubles per lattice row, in LDS); the suffixes arrive from the right neighbour exactly like the
// suffix sums of dM * U do.  Only the gradient of the REGISTER-RESIDENT side (y) is produced -- per-lane accumulators, no
// cross-lane traffic; the other side comes from a second launch with the roles exchanged (or, for a symmetric Gram, from
// the symmetry k(x, y) = k(y, x): upstream G + G^T).
// =====================================================================================================================
template <int C, int LQ>
struct WaveUndo {
    double qf[LQ][C], qfg[LQ];                  // Q_m[a][b_c] and the ghost column Q_m[a][b_0 - 1]; undone row by row
    double qb[LQ][C], qbg[LQ];                  // as WaveBwd
    double svout[LQ], sufout[LQ];

    GPSIG_HD void init(const WaveFwd<C, LQ>& fw) {
#pragma unroll
        for (int m = 0; m < LQ; ++m) {
            qfg[m] = fw.qg[m];
            qbg[m] = svout[m] = sufout[m] = 0.0;
#pragma unroll
            for (int c = 0; c < C; ++c) { qf[m][c] = fw.q[m][c]; qb[m][c] = 0.0; }
        }
    }
    // rowtot[m-1] = sum_b R_m[a][b]; sufin[m-1] / svin[p-1]: the right neighbour's sufout / svout of ITS previous step.
    // first_row: a == 0 (Q[-1][.] == 0 exactly); first_lane: b_0 == 0 (Q[.][-1] == 0 exactly).
    GPSIG_HD void step(const double (&dm)[C], const double (&clev)[LQ + 2], const double (&rowtot)[LQ], const double (&sufin)[LQ],
                       const double (&svin)[LQ], int M, bool first_row, bool first_lane, double (&lam)[C]) {
        double D[LQ + 1][C];                    // D[m][c] = Q_m[a-1][b_c - 1];  D[0] == 1
#pragma unroll
        for (int c = 0; c < C; ++c) D[0][c] = 1.0;
#pragma unroll
        for (int m = 1; m <= LQ; ++m) {
            if (m < M) {
                double v = sufin[m - 1] - rowtot[m - 1];
#pragma unroll
                for (int c = C - 1; c >= 0; --c) {
                    qf[m - 1][c] += v;                              // Q_m[a-1][b_c] = Q_m[a][b_c] - (rowtot - suffix beyond b_c)
                    v = fma(dm[c], D[m - 1][c], v);                 // + R_m[a][b_c]
                }
                qfg[m - 1] += v;
                sufout[m - 1] = v + rowtot[m - 1];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                double d = c == 0 ? qfg[m - 1] : qf[m - 1][c > 0 ? c - 1 : 0];
                if (first_row || (first_lane && c == 0) || m >= M) d = 0.0;
                D[m][c] = d;
            }
        }
        double U[LQ + 2][C];
#pragma unroll
        for (int p = 1; p <= LQ + 1; ++p)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int pi = p - 1 < LQ ? p - 1 : LQ - 1;
                if (p < M) U[p][c] = clev[p] + (c < C - 1 ? qb[pi][c < C - 1 ? c + 1 : c] : qbg[pi]);
                else U[p][c] = p == M ? clev[p] : 0.0;
            }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            double l = U[1][c];
#pragma unroll
            for (int p = 2; p <= LQ + 1; ++p)
                if (p <= M) l = fma(D[p - 1][c], U[p][c], l);
            lam[c] = l;
        }
#pragma unroll
        for (int p = 1; p <= LQ; ++p)
            if (p < M) {
                double sv = svin[p - 1];
#pragma unroll
                for (int c = C - 1; c >= 0; --c) {
                    sv = fma(dm[c], U[p + 1][c], sv);
                    qb[p - 1][c] += sv;
                }
                svout[p - 1] = sv;
                qbg[p - 1] += svin[p - 1];
            }
    }
};

// Backward-sweep dM generator that also accumulates the gradient of the lane's own y points.
//   g[c][f], c = 0..C: gradient with respect to feature f of point y_{b_0 + c}   (MODE_PT_NODIFF: c < C)
// MODE_INC works on increments (dM = <dx_a, dy_b>): g is first accumulated per increment column and folded to points by fold().
template <int C, int DP, int MODE>
struct WaveGy {
    double y[C + 1][DP], ys[C + 1];     // the lane's points (MODE_INC: y[c], c < C, holds the increment y_{b+1} - y_b)
    double g[C + 1][DP];
    double xk[DP];                       // the x point row kept from the previous step
    double rd[C];                        // MODE_PT_DIFF: kappa(x_{a+1}, y_{b+1}) - kappa(x_{a+1}, y_b)
    double wy[C + 1], wx[C + 1];         // MODE_PT_DIFF: d kappa(x_{a+1}, y_c)/dy = wx * x + wy * y   (coefficients of the kept row)
    double lamk[C];                      // Lam of the row processed in the previous step
    double gp0;
    int nvalid;

    // once per run of x sequences: the lane's points and zeroed accumulators
    GPSIG_HD void set_y(const double (&ypts)[C + 1][DP], int nvalid_) {
        nvalid = nvalid_;
        gp0 = 0.0;
#pragma unroll
        for (int c = 0; c <= C; ++c) {
            double s = 0.0;
#pragma unroll
            for (int f = 0; f < DP; ++f) {
                if (MODE == MODE_INC) { if (c < C) y[c][f] = ypts[c + 1][f] - ypts[c][f]; else y[c][f] = 0.0; }
                else y[c][f] = ypts[c][f];
                s = fma(ypts[c][f], ypts[c][f], s);
                g[c][f] = 0.0;
            }
            ys[c] = s;
        }
    }
    // gradient of point c (0 .. nvalid, or .. nvalid-1 without differences), feature f, after a run
    GPSIG_HD double point_grad(int c, int f) const {
        if (MODE != MODE_INC) return g[c][f];
        return (c > 0 ? g[c > 0 ? c - 1 : 0][f] : 0.0) - (c < nvalid ? g[c < C ? c : C - 1][f] : 0.0);
    }
    // kappa and d kappa/dy coefficients of one x row against the lane's points
    GPSIG_HD void eval_row(const double (&x)[DP], int kind, double p0, double p1, double (&k)[C + 1], double (&cwy)[C + 1], double (&cwx)[C + 1],
                           double (&dp)[C + 1], int npts) const {
        double xs = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f) xs = fma(x[f], x[f], xs);
#pragma unroll
        for (int c = 0; c <= C; ++c) {
            if (c < npts) {
                double in = 0.0;
#pragma unroll
                for (int f = 0; f < DP; ++f) in = fma(x[f], y[c][f], in);
                const BaseGrad bg = base_eval_grad(kind, in, xs, ys[c], p0, p1);
                k[c] = bg.k;
                cwx[c] = bg.cy - bg.cd;          // d kappa/dy = (cy - cd) x + (cx2 + cd) y
                cwy[c] = bg.cx2 + bg.cd;
                dp[c] = bg.dp0;
            } else {
                k[c] = cwx[c] = cwy[c] = dp[c] = 0.0;
            }
        }
    }
    // ---- forward sweep (dM only, no derivatives): same storage as the backward sweep uses
    GPSIG_HD void prime_fwd(const double (&x0)[DP], int kind, double p0, double p1) {
#pragma unroll
        for (int f = 0; f < DP; ++f) xk[f] = x0[f];
        if (MODE == MODE_PT_DIFF) {
            double k[C + 1];
            kappa_row(x0, kind, p0, p1, k, nvalid + 1);
#pragma unroll
            for (int c = 0; c < C; ++c) rd[c] = k[c + 1] - k[c];
        }
    }
    GPSIG_HD void kappa_row(const double (&x)[DP], int kind, double p0, double p1, double (&k)[C + 1], int npts) const {
        double xs = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f) xs = fma(x[f], x[f], xs);
#pragma unroll
        for (int c = 0; c <= C; ++c) {
            double in = 0.0;
#pragma unroll
            for (int f = 0; f < DP; ++f) in = fma(x[f], y[c][f], in);
            k[c] = c < npts ? base_eval<double>(kind, in, xs, ys[c], p0, p1) : 0.0;
        }
    }
    // xnew: x_{a+1} (difference modes) or x_a
    GPSIG_HD void row_fwd(const double (&xnew)[DP], int kind, double p0, double p1, double (&dm)[C]) {
        if (MODE == MODE_INC) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int f = 0; f < DP; ++f) acc = fma(xnew[f] - xk[f], y[c][f], acc);
                dm[c] = acc;
            }
#pragma unroll
            for (int f = 0; f < DP; ++f) xk[f] = xnew[f];
        } else if (MODE == MODE_PT_DIFF) {
            double k[C + 1];
            kappa_row(xnew, kind, p0, p1, k, nvalid + 1);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double nd = k[c + 1] - k[c];
                dm[c] = nd - rd[c];
                rd[c] = nd;
            }
        } else {
            double k[C + 1];
            kappa_row(xnew, kind, p0, p1, k, nvalid);
#pragma unroll
            for (int c = 0; c < C; ++c) dm[c] = k[c];
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (c >= nvalid) dm[c] = 0.0;
    }

    // before the backward sweep: x_{R1} (difference modes)
    GPSIG_HD void prime(const double (&x)[DP], int kind, double p0, double p1) {
#pragma unroll
        for (int c = 0; c < C; ++c) lamk[c] = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f) xk[f] = x[f];
        if (MODE == MODE_PT_DIFF) {
            double k[C + 1], a1[C + 1], a2[C + 1], dp[C + 1];
            eval_row(x, kind, p0, p1, k, a1, a2, dp, nvalid + 1);
#pragma unroll
            for (int c = 0; c < C; ++c) rd[c] = k[c + 1] - k[c];
#pragma unroll
            for (int c = 0; c <= C; ++c) { wy[c] = a1[c]; wx[c] = a2[c]; dpk[c] = dp[c]; }
        }
    }
    double dpk[C + 1];                   // d kappa(x_kept, y_c) / d base_params[0]

    // dM of lattice row a from x_a (backward order); keeps what the contraction of this row needs
    double xcur[DP], kwy[C + 1], kwx[C + 1], kdp[C + 1], dxa[DP];
    GPSIG_HD void row(const double (&xa)[DP], int kind, double p0, double p1, double (&dm)[C]) {
        if (MODE == MODE_INC) {
#pragma unroll
            for (int f = 0; f < DP; ++f) dxa[f] = xk[f] - xa[f];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int f = 0; f < DP; ++f) acc = fma(dxa[f], y[c][f], acc);
                dm[c] = acc;
            }
        } else if (MODE == MODE_PT_DIFF) {
            double k[C + 1], a1[C + 1], a2[C + 1], dp[C + 1];
            eval_row(xa, kind, p0, p1, k, a1, a2, dp, nvalid + 1);
#pragma unroll
            for (int c = 0; c <= C; ++c) { kwy[c] = a1[c]; kwx[c] = a2[c]; kdp[c] = dp[c]; }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double nd = k[c + 1] - k[c];
                dm[c] = rd[c] - nd;
                rd[c] = nd;
            }
        } else {
            double k[C + 1], a1[C + 1], a2[C + 1], dp[C + 1];
            eval_row(xa, kind, p0, p1, k, a1, a2, dp, nvalid);
#pragma unroll
            for (int c = 0; c <= C; ++c) { kwy[c] = a1[c]; kwx[c] = a2[c]; kdp[c] = dp[c]; }
#pragma unroll
            for (int c = 0; c < C; ++c) dm[c] = k[c];
        }
#pragma unroll
        for (int f = 0; f < DP; ++f) xcur[f] = xa[f];
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (c >= nvalid) dm[c] = 0.0;
    }
    // after Lam of row a is known
    GPSIG_HD void contract(const double (&lam_in)[C]) {
        double lam[C];
#pragma unroll
        for (int c = 0; c < C; ++c) lam[c] = c < nvalid ? lam_in[c] : 0.0;
        if (MODE == MODE_INC) {
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int f = 0; f < DP; ++f) g[c][f] = fma(lam[c], dxa[f], g[c][f]);
        } else if (MODE == MODE_PT_DIFF) {
            // adjoint of rd of the KEPT row (x_{a+1}): grd[c] = Lam[a][c] - Lam[a+1][c]; point c receives grd[c-1] - grd[c]
            double h[C + 1];
#pragma unroll
            for (int c = 0; c <= C; ++c) {
                const double left = c > 0 ? lam[c - 1] - lamk[c - 1] : 0.0;
                const double right = c < C ? lam[c] - lamk[c] : 0.0;
                h[c] = left - right;
                gp0 = fma(h[c], dpk[c], gp0);
            }
#pragma unroll
            for (int c = 0; c <= C; ++c) {
                const double a = h[c] * wx[c], b = h[c] * wy[c];
#pragma unroll
                for (int f = 0; f < DP; ++f) g[c][f] = fma(a, xk[f], fma(b, y[c][f], g[c][f]));
            }
#pragma unroll
            for (int c = 0; c <= C; ++c) { wy[c] = kwy[c]; wx[c] = kwx[c]; dpk[c] = kdp[c]; }
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double a = lam[c] * kwx[c], b = lam[c] * kwy[c];
                gp0 = fma(lam[c], kdp[c], gp0);
#pragma unroll
                for (int f = 0; f < DP; ++f) g[c][f] = fma(a, xcur[f], fma(b, y[c][f], g[c][f]));
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) lamk[c] = lam[c];
#pragma unroll
        for (int f = 0; f < DP; ++f) xk[f] = xcur[f];
    }
    // after the last row (a == 0) of a pair: the kept row is x_0 (MODE_PT_DIFF only)
    GPSIG_HD void finish_pair() {
        if (MODE == MODE_PT_DIFF) {
            double h[C + 1];
#pragma unroll
            for (int c = 0; c <= C; ++c) {
                const double left = c > 0 ? -lamk[c - 1] : 0.0;
                const double right = c < C ? -lamk[c] : 0.0;
                h[c] = left - right;
                gp0 = fma(h[c], dpk[c], gp0);
            }
#pragma unroll
            for (int c = 0; c <= C; ++c) {
                const double a = h[c] * wx[c], b = h[c] * wy[c];
#pragma unroll
                for (int f = 0; f < DP; ++f) g[c][f] = fma(a, xk[f], fma(b, y[c][f], g[c][f]));
            }
        }
    }
};

}  // namespace gpsig
