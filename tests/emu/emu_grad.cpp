// emu_grad.cpp -- CPU harness around gpsig_amd/csrc/grad_core.hpp (TEST INFRASTRUCTURE).
// Runs the very per-pair code the gfx950 gradient kernels run, one pair after the other, with plain adds in place
// of atomics, so the adjoint recursions can be checked against torch.autograd on a machine without a GPU.
#include <cstdint>
#include <cstring>
#include <array>
#include <algorithm>
#include <vector>

#include "grad_core.hpp"
#include "grad_wave_core.hpp"

using namespace gpsig;

namespace {
int pad_of(int d) { return d <= 4 ? 4 : (d <= 8 ? 8 : (d <= 16 ? 16 : 32)); }

std::vector<double> timemajor(const double* X, int N, int L, int d, int DP) {
    std::vector<double> T(size_t(L) * DP * N, 0.0);
    for (int i = 0; i < N; ++i)
        for (int t = 0; t < L; ++t)
            for (int f = 0; f < d; ++f) T[(size_t(t) * DP + f) * N + i] = X[(size_t(i) * L + t) * d + f];
    return T;
}
void from_timemajor(const std::vector<double>& T, double* X, int N, int L, int d, int DP) {
    for (int i = 0; i < N; ++i)
        for (int t = 0; t < L; ++t)
            for (int f = 0; f < d; ++f) X[(size_t(i) * L + t) * d + f] = T[(size_t(t) * DP + f) * N + i];
}
std::vector<double> pad_rows(const double* Z, int64_t rows, int d, int DP) {
    std::vector<double> P(size_t(rows) * DP, 0.0);
    for (int64_t r = 0; r < rows; ++r)
        for (int f = 0; f < d; ++f) P[r * DP + f] = Z[r * d + f];
    return P;
}

template <int DP>
void seq_run(SeqGradArgs& A, int N1, int N2, double* levels_out, int M) {
    std::vector<double> lev(M + 1);
    for (int i = 0; i < N1; ++i) {
        const int jlo = A.diag ? i : 0, jhi = A.diag ? i + 1 : N2;
        for (int j = jlo; j < jhi; ++j) {
            A.pairs = 1;
            A.levels = lev.data();
            SeqPairGrad<DP> P(A, i, j, 0, true);
            P.forward();
            P.backward();
            P.contract();
            if (levels_out)
                for (int m = 0; m <= M; ++m) levels_out[A.diag ? size_t(m) * N1 + i : (size_t(m) * N1 + i) * N2 + j] = lev[m];
        }
    }
}
template <int DP>
void tvs_run(TvsGradArgs& A, double* levels_out) {
    std::vector<double> lev(A.M + 1);
    for (int t = 0; t < A.T; ++t)
        for (int n = 0; n < A.N; ++n) {
            A.pairs = 1;
            A.levels = lev.data();
            TvsPairGrad<DP> P(A, t, n, 0, true);
            P.forward();
            P.backward();
            P.contract();
            if (levels_out)
                for (int m = 0; m <= A.M; ++m) levels_out[(size_t(m) * A.T + t) * A.N + n] = lev[m];
        }
}
template <int DP, int E>
void tvs_run_fused(TvsGradArgs& A, double* levels_out) {
    A.levels = levels_out;
    A.pairs = int64_t(A.T) * A.N; A.level_t = A.N; A.level_n = 1;
    for (int t = 0; t < A.T; ++t)
        for (int n = 0; n < A.N; ++n) TvsPairGradFused<DP, 4, E>(A, t, n, true).run();
}
template <int DP>
void tens_run(TensGradArgs& A) {
    for (int t = 0; t < A.T; ++t)
        for (int t2 = 0; t2 < A.T; ++t2) TensPairGrad<DP>(A, t, t2, true).run();
}
}  // namespace

extern "C" {

// X (N1, L1, d), Y (N2, L2, d) or NULL (symmetric: Y = X, both roles accumulate into gX); diag: pairs (i, i).
// G (M+1, N1, N2) or (M+1, N1) when diag.  Outputs are overwritten.
int emu_seq_grad(const double* X, const double* Y, int N1, int N2, int L1, int L2, int d, int M, int kind, int mode, double p0, double p1,
                 int diag, const double* G, double* gX, double* gY, double* levels_out, double* gbase) {
    const int DP = pad_of(d);
    const bool sym = !diag && !Y;
    if (diag || sym) { N2 = N1; L2 = L1; }
    std::vector<double> xT = timemajor(X, N1, L1, d, DP), yT, gxT(xT.size(), 0.0), gyT;
    if (!diag && !sym) { yT = timemajor(Y, N2, L2, d, DP); gyT.assign(yT.size(), 0.0); }
    const int dr = mode == MODE_PT_NODIFF ? 0 : 1;
    std::vector<double> scr(size_t(M) * (L1 - dr) * (L2 - dr) + 8);
    double gb[2] = {0, 0};
    SeqGradArgs A;
    memset(&A, 0, sizeof(A));
    A.xT = xT.data(); A.yT = (diag || sym) ? xT.data() : yT.data();
    A.gxT = gxT.data(); A.gyT = (diag || sym) ? gxT.data() : gyT.data();
    A.xstride = N1; A.ystride = N2;
    A.N1 = N1; A.N2 = N2; A.L1 = L1; A.L2 = L2; A.M = M; A.kind = kind; A.mode = mode; A.p0 = p0; A.p1 = p1;
    A.diag = diag;
    A.G = G; A.gm = diag ? N1 : int64_t(N1) * N2; A.gi = diag ? 1 : N2; A.gj = diag ? 0 : 1;
    A.scratch = scr.data();
    A.gbase = gb;
    switch (DP) {
        case 4: seq_run<4>(A, N1, N2, levels_out, M); break;
        case 8: seq_run<8>(A, N1, N2, levels_out, M); break;
        case 16: seq_run<16>(A, N1, N2, levels_out, M); break;
        default: seq_run<32>(A, N1, N2, levels_out, M); break;
    }
    from_timemajor(gxT, gX, N1, L1, d, DP);
    if (!diag && !sym) from_timemajor(gyT, gY, N2, L2, d, DP);
    if (gbase) { gbase[0] = gb[0]; gbase[1] = gb[1]; }
    return 0;
}

// Z (lt, T, [2,] d), X (N, L, d), G (M+1, T, N)
int emu_tvs_grad(const double* Z, const double* X, int T, int N, int L, int d, int M, int kind, int incr, int diff, double p0, double p1,
                 const double* G, double* gZ, double* gX, double* levels_out, double* gbase, int fused) {
    const int DP = pad_of(d), lt = M * (M + 1) / 2, E = incr ? 2 : 1;
    const int64_t rows = int64_t(lt) * T * E;
    std::vector<double> zp = pad_rows(Z, rows, d, DP), gzp(zp.size(), 0.0), xT = timemajor(X, N, L, d, DP), gxT(xT.size(), 0.0);
    const int R = diff ? L - 1 : L;
    std::vector<double> scr(size_t(lt + M * (M - 1) / 2) * R + 8);
    double gb[2] = {0, 0};
    TvsGradArgs A;
    memset(&A, 0, sizeof(A));
    A.z = zp.data(); A.gz = gzp.data(); A.xT = xT.data(); A.gxT = gxT.data(); A.xstride = N;
    A.T = T; A.N = N; A.L = L; A.M = M; A.kind = kind; A.incr = incr; A.diff = diff; A.p0 = p0; A.p1 = p1;
    A.G = G; A.gm = int64_t(T) * N; A.gt = N; A.gn = 1;
    A.scratch = scr.data();
    A.gbase = gb;
    if (fused) {
        if (M > 4 || DP > 8) return -2;
        if (DP == 4 && E == 1) tvs_run_fused<4, 1>(A, levels_out);
        else if (DP == 4) tvs_run_fused<4, 2>(A, levels_out);
        else if (E == 1) tvs_run_fused<8, 1>(A, levels_out);
        else tvs_run_fused<8, 2>(A, levels_out);
    } else switch (DP) {
        case 4: tvs_run<4>(A, levels_out); break;
        case 8: tvs_run<8>(A, levels_out); break;
        case 16: tvs_run<16>(A, levels_out); break;
        default: tvs_run<32>(A, levels_out); break;
    }
    for (int64_t r = 0; r < rows; ++r)
        for (int f = 0; f < d; ++f) gZ[r * d + f] = gzp[r * DP + f];
    from_timemajor(gxT, gX, N, L, d, DP);
    if (gbase) { gbase[0] = gb[0]; gbase[1] = gb[1]; }
    return 0;
}

// Z (lt, T, [2,] d), G (M+1, T, T)
int emu_tens_grad(const double* Z, int T, int d, int M, int kind, int incr, double p0, double p1, const double* G, double* gZ, double* gbase,
                  int row_owned) {
    const int DP = pad_of(d), lt = M * (M + 1) / 2, E = incr ? 2 : 1;
    const int64_t rows = int64_t(lt) * T * E;
    std::vector<double> zp = pad_rows(Z, rows, d, DP), gzp(zp.size(), 0.0);
    double gb[2] = {0, 0};
    TensGradArgs A;
    memset(&A, 0, sizeof(A));
    A.z = zp.data(); A.gz = gzp.data(); A.T = T; A.M = M; A.kind = kind; A.incr = incr; A.p0 = p0; A.p1 = p1;
    A.G = G; A.gm = int64_t(T) * T; A.gt = T; A.gn = 1;
    A.gbase = gb;
    if (row_owned) {
        for (int sl = 0; sl < 3; ++sl)
            for (int t = 0; t < T; ++t) {
                if (DP == 4 && E == 1) TensRowGrad<4, 1>(A, t, true).run(sl, 3);
                else if (DP == 4) TensRowGrad<4, 2>(A, t, true).run(sl, 3);
                else if (DP == 8 && E == 1) TensRowGrad<8, 1>(A, t, true).run(sl, 3);
                else if (DP == 8) TensRowGrad<8, 2>(A, t, true).run(sl, 3);
                else if (DP == 16 && E == 1) TensRowGrad<16, 1>(A, t, true).run(sl, 3);
                else if (DP == 16) TensRowGrad<16, 2>(A, t, true).run(sl, 3);
                else if (E == 1) TensRowGrad<32, 1>(A, t, true).run(sl, 3);
                else TensRowGrad<32, 2>(A, t, true).run(sl, 3);
            }
    } else switch (DP) {
        case 4: tens_run<4>(A); break;
        case 8: tens_run<8>(A); break;
        case 16: tens_run<16>(A); break;
        default: tens_run<32>(A); break;
    }
    for (int64_t r = 0; r < rows; ++r)
        for (int f = 0; f < d; ++f) gZ[r * d + f] = gzp[r * DP + f];
    if (gbase) { gbase[0] = gb[0]; gbase[1] = gb[1]; }
    return 0;
}

}  // extern "C"

namespace {
// lock-step emulation of one pair group of the wave kernel: G lanes, C columns each
template <int G, int C, int DP, int LQ, int MODE>
void wave_pair(const double* X, const double* Y, int i, int j, int L1, int L2, int d, int M, int kind, double p0, double p1,
               const double* clev_in /* M+1 */, std::vector<double>& lam_out) {
    const int dr = MODE == MODE_PT_NODIFF ? 0 : 1;
    const int R1 = L1 - dr, R2 = L2 - dr, TF = R1 + G - 1;
    auto load = [&](const double* S, int seq, int L, int r, double (&v)[DP]) {
        for (int f = 0; f < DP; ++f) v[f] = (r >= 0 && r < L && f < d) ? S[(size_t(seq) * L + r) * d + f] : 0.0;
    };
    std::vector<WaveDm<C, DP, MODE>> dm(G);
    std::vector<double> scr(size_t(M > 1 ? M - 1 : 1) * TF * G * C, 0.0);
    auto slot = [&](int m, int tf, int l, int c) -> double& { return scr[((size_t(m) * TF + tf) * G + l) * C + c]; };
    double clev[LQ + 2];
    for (int p = 0; p < LQ + 2; ++p) clev[p] = (p >= 1 && p <= M) ? clev_in[p] : 0.0;
    for (int l = 0; l < G; ++l) {
        double ypts[C + 1][DP];
        for (int c = 0; c <= C; ++c) load(Y, j, L2, C * l + c, ypts[c]);
        int nv = R2 - C * l;
        dm[l].set_y(ypts, nv < 0 ? 0 : (nv > C ? C : nv));
    }
    lam_out.assign(size_t(R1) * R2, 0.0);
    {   // forward
        std::vector<WaveFwd<C, LQ>> fw(G);
        for (int l = 0; l < G; ++l) {
            fw[l].reset();
            if (MODE != MODE_PT_NODIFF) { double x0[DP]; load(X, i, L1, 0, x0); dm[l].prime(x0, kind, p0, p1); }
        }
        for (int t = 0; t < TF; ++t) {
            std::vector<std::array<double, LQ + 2>> snap(G);
            for (int l = 0; l < G; ++l)
                for (int m = 0; m < LQ + 2; ++m) snap[l][m] = l > 0 ? fw[l - 1].sout[m] : 0.0;
            for (int l = 0; l < G; ++l) {
                const int a = t - l;
                if (a < 0 || a >= R1) continue;
                double cin[LQ + 2], xn[DP], dmv[C];
                for (int m = 0; m < LQ + 2; ++m) cin[m] = snap[l][m];
                cin[0] = 0.0;
                load(X, i, L1, a + dr, xn);
                dm[l].row(xn, true, kind, p0, p1, dmv);
                fw[l].step(dmv, cin, M);
                for (int m = 0; m < LQ; ++m)
                    if (m < M - 1)
                        for (int c = 0; c < C; ++c) slot(m, t, l, c) = fw[l].q[m][c];
            }
        }
    }
    {   // backward
        std::vector<WaveBwd<C, LQ>> bw(G);
        for (int l = 0; l < G; ++l) {
            bw[l].reset();
            if (MODE != MODE_PT_NODIFF) { double xl[DP]; load(X, i, L1, R1, xl); dm[l].prime(xl, kind, p0, p1); }
        }
        for (int u = 0; u < TF; ++u) {
            std::vector<std::array<double, LQ>> snap(G);
            for (int l = 0; l < G; ++l)
                for (int p = 0; p < LQ; ++p) snap[l][p] = l < G - 1 ? bw[l + 1].svout[p] : 0.0;
            for (int l = 0; l < G; ++l) {
                const int a = R1 - 1 - (u - (G - 1 - l));
                if (a < 0 || a >= R1) continue;
                double sin[LQ], xn[DP], dmv[C], qfd[LQ][C], lv[C];
                for (int p = 0; p < LQ; ++p) sin[p] = snap[l][p];
                load(X, i, L1, a, xn);
                dm[l].row(xn, false, kind, p0, p1, dmv);
                const int tf = a - 1 + l;
                for (int m = 0; m < LQ; ++m)
                    for (int c = 0; c < C; ++c) {
                        double v = 0.0;
                        if (m < M - 1 && a > 0) {
                            if (c > 0) v = slot(m, tf, l, c - 1);
                            else if (l > 0) v = slot(m, tf - 1, l - 1, C - 1);
                        }
                        qfd[m][c] = v;
                    }
                bw[l].step(dmv, clev, qfd, sin, M, lv);
                for (int c = 0; c < C; ++c)
                    if (c < dm[l].nvalid) lam_out[size_t(a) * R2 + C * l + c] = lv[c];
            }
        }
    }
}

// Lam of one pair through the scratch-free formulation: forward sweep keeping the row totals, backward sweep undoing the
// forward recursion (WaveUndo) with the plain dM generator (WaveDm) -- the sweeps of seq_lam_undo_kernel.
template <int G, int C, int DP, int LQ, int MODE>
void wave_pair_undo(const double* X, const double* Y, int i, int j, int L1, int L2, int d, int M, int kind, double p0, double p1,
                    const double* clev_in, std::vector<double>& lam_out) {
    const int dr = MODE == MODE_PT_NODIFF ? 0 : 1;
    const int R1 = L1 - dr, R2 = L2 - dr, TF = R1 + G - 1;
    auto load = [&](const double* S, int seq, int L, int r, double (&v)[DP]) {
        for (int f = 0; f < DP; ++f) v[f] = (r >= 0 && r < L && f < d) ? S[(size_t(seq) * L + r) * d + f] : 0.0;
    };
    std::vector<WaveDm<C, DP, MODE>> dm(G);
    std::vector<WaveFwd<C, LQ>> fw(G);
    std::vector<double> rowtot(size_t(R1 > 0 ? R1 : 1) * LQ, 0.0);
    double clev[LQ + 2];
    for (int p = 0; p < LQ + 2; ++p) clev[p] = (p >= 1 && p <= M) ? clev_in[p] : 0.0;
    const int last_lane = R2 > 0 ? (R2 - 1) / C : 0;
    lam_out.assign(size_t(R1) * R2, 0.0);
    for (int l = 0; l < G; ++l) {
        double ypts[C + 1][DP];
        for (int c = 0; c <= C; ++c) load(Y, j, L2, C * l + c, ypts[c]);
        int nv = R2 - C * l;
        dm[l].set_y(ypts, nv < 0 ? 0 : (nv > C ? C : nv));
        fw[l].reset();
        if (MODE != MODE_PT_NODIFF) { double x0[DP]; load(X, i, L1, 0, x0); dm[l].prime(x0, kind, p0, p1); }
    }
    for (int t = 0; t < TF; ++t) {
        std::vector<std::array<double, LQ + 2>> snap(G);
        for (int l = 0; l < G; ++l)
            for (int m = 0; m < LQ + 2; ++m) snap[l][m] = l > 0 ? fw[l - 1].sout[m] : 0.0;
        for (int l = 0; l < G; ++l) {
            const int a = t - l;
            if (a < 0 || a >= R1) continue;
            double cin[LQ + 2], xn[DP], dmv[C];
            for (int m = 0; m < LQ + 2; ++m) cin[m] = snap[l][m];
            cin[0] = 0.0;
            load(X, i, L1, a + dr, xn);
            dm[l].row(xn, true, kind, p0, p1, dmv);
            fw[l].step(dmv, cin, M);
            if (l == last_lane)
                for (int m = 1; m <= LQ; ++m) rowtot[size_t(a) * LQ + m - 1] = m < M ? fw[l].sout[m] : 0.0;
        }
    }
    std::vector<WaveUndo<C, LQ>> bw(G);
    for (int l = 0; l < G; ++l) {
        bw[l].init(fw[l]);
        if (MODE != MODE_PT_NODIFF) { double xl[DP]; load(X, i, L1, R1, xl); dm[l].prime(xl, kind, p0, p1); }
    }
    for (int u = 0; u < TF; ++u) {
        std::vector<std::array<double, 2 * LQ>> snap(G);
        for (int l = 0; l < G; ++l)
            for (int p = 0; p < LQ; ++p) {
                snap[l][p] = l < G - 1 ? bw[l + 1].sufout[p] : 0.0;
                snap[l][LQ + p] = l < G - 1 ? bw[l + 1].svout[p] : 0.0;
            }
        for (int l = 0; l < G; ++l) {
            const int a = R1 - 1 - (u - (G - 1 - l));
            if (a < 0 || a >= R1) continue;
            double sufin[LQ], svin[LQ], rt[LQ], xn[DP], dmv[C], lv[C];
            for (int p = 0; p < LQ; ++p) { sufin[p] = snap[l][p]; svin[p] = snap[l][LQ + p]; rt[p] = rowtot[size_t(a) * LQ + p]; }
            load(X, i, L1, a, xn);
            dm[l].row(xn, false, kind, p0, p1, dmv);
            bw[l].step(dmv, clev, rt, sufin, svin, M, a == 0, l == 0, lv);
            for (int c = 0; c < C; ++c)
                if (c < dm[l].nvalid) lam_out[size_t(a) * R2 + C * l + c] = lv[c];
        }
    }
}

template <int G, int C, int DP, int LQ>
void wave_pair_mode(int mode, const double* X, const double* Y, int i, int j, int L1, int L2, int d, int M, int kind, double p0, double p1,
                    const double* clev, std::vector<double>& lam, bool undo) {
    if (undo) {
        if (mode == MODE_INC) wave_pair_undo<G, C, DP, LQ, MODE_INC>(X, Y, i, j, L1, L2, d, M, kind, p0, p1, clev, lam);
        else if (mode == MODE_PT_DIFF) wave_pair_undo<G, C, DP, LQ, MODE_PT_DIFF>(X, Y, i, j, L1, L2, d, M, kind, p0, p1, clev, lam);
        else wave_pair_undo<G, C, DP, LQ, MODE_PT_NODIFF>(X, Y, i, j, L1, L2, d, M, kind, p0, p1, clev, lam);
        return;
    }
    if (mode == MODE_INC) wave_pair<G, C, DP, LQ, MODE_INC>(X, Y, i, j, L1, L2, d, M, kind, p0, p1, clev, lam);
    else if (mode == MODE_PT_DIFF) wave_pair<G, C, DP, LQ, MODE_PT_DIFF>(X, Y, i, j, L1, L2, d, M, kind, p0, p1, clev, lam);
    else wave_pair<G, C, DP, LQ, MODE_PT_NODIFF>(X, Y, i, j, L1, L2, d, M, kind, p0, p1, clev, lam);
}
}  // namespace

extern "C" {
// The wave formulation (grad_wave_core.hpp) for the lattice sweeps, followed by the per-pair contraction of grad_core.hpp.
// Same contract as emu_seq_grad.  (Gg, Cc) in {(16,2), (16,4), (64,2)}; returns -2 for anything else or if the lattice does not fit.
// undo: Lam from the scratch-free sweeps (wave_pair_undo) instead of the stored forward lattice.
int emu_seq_grad_wave(const double* X, const double* Y, int N1, int N2, int L1, int L2, int d, int M, int kind, int mode, double p0, double p1,
                      int diag, const double* G, double* gX, double* gY, double* gbase, int Gg, int Cc, int undo) {
    const int DP = pad_of(d);
    const bool sym = !diag && !Y;
    if (diag || sym) { N2 = N1; L2 = L1; Y = X; }
    const int dr = mode == MODE_PT_NODIFF ? 0 : 1;
    const int R1 = L1 - dr, R2 = L2 - dr;
    if (DP != 4 && DP != 8) return -2;
    if (R2 > Gg * Cc || M > 8) return -2;
    std::vector<double> xT = timemajor(X, N1, L1, d, DP), yT, gxT(xT.size(), 0.0), gyT;
    const bool two = !diag && !sym;
    if (two) { yT = timemajor(Y, N2, L2, d, DP); gyT.assign(yT.size(), 0.0); }
    double gb[2] = {0, 0};
    SeqGradArgs A;
    memset(&A, 0, sizeof(A));
    A.xT = xT.data(); A.yT = two ? yT.data() : xT.data();
    A.gxT = gxT.data(); A.gyT = two ? gyT.data() : gxT.data();
    A.xstride = N1; A.ystride = N2;
    A.N1 = N1; A.N2 = N2; A.L1 = L1; A.L2 = L2; A.M = M; A.kind = kind; A.mode = mode; A.p0 = p0; A.p1 = p1;
    A.diag = diag; A.pairs = 1; A.gbase = gb;
    std::vector<double> lam, clev(M + 1);
    for (int i = 0; i < N1; ++i)
        for (int j = diag ? i : 0; j < (diag ? i + 1 : N2); ++j) {
            for (int m = 0; m <= M; ++m) clev[m] = diag ? G[size_t(m) * N1 + i] : G[(size_t(m) * N1 + i) * N2 + j];
#define WP(GG, CC, DD, LL) wave_pair_mode<GG, CC, DD, LL>(mode, X, Y, i, j, L1, L2, d, M, kind, p0, p1, clev.data(), lam, undo != 0)
            const bool small = M <= 5;
            if (Gg == 16 && Cc == 2) { if (DP == 4) { if (small) WP(16, 2, 4, 4); else WP(16, 2, 4, 7); } else { if (small) WP(16, 2, 8, 4); else WP(16, 2, 8, 7); } }
            else if (Gg == 16 && Cc == 4) { if (DP == 4) { if (small) WP(16, 4, 4, 4); else WP(16, 4, 4, 7); } else { if (small) WP(16, 4, 8, 4); else WP(16, 4, 8, 7); } }
            else if (Gg == 64 && Cc == 2) { if (DP == 4) { if (small) WP(64, 2, 4, 4); else WP(64, 2, 4, 7); } else { if (small) WP(64, 2, 8, 4); else WP(64, 2, 8, 7); } }
            else return -2;
#undef WP
            std::vector<double> scr(size_t(M) * (R1 > 0 ? R1 : 0) * (R2 > 0 ? R2 : 0) + 8, 0.0);
            std::copy(lam.begin(), lam.end(), scr.begin());           // slot 0 = Lam
            A.scratch = scr.data();
            auto run = [&](auto P) { P.contract(); };
            if (DP == 4) run(SeqPairGrad<4>(A, i, j, 0, true)); else run(SeqPairGrad<8>(A, i, j, 0, true));
        }
    from_timemajor(gxT, gX, N1, L1, d, DP);
    if (two) from_timemajor(gyT, gY, N2, L2, d, DP);
    if (gbase) { gbase[0] = gb[0]; gbase[1] = gb[1]; }
    return 0;
}
}

namespace {
// lock-step emulation of the scratch-free wave formulation: gradient of the register-resident side only.
// gy: (L2, d) accumulated.
template <int G, int C, int DP, int LQ, int MODE>
double wave2_pair(const double* X, const double* Y, int i, int j, int L1, int L2, int d, int M, int kind, double p0, double p1,
                  const double* clev_in, double* gy) {
    const int dr = MODE == MODE_PT_NODIFF ? 0 : 1;
    const int R1 = L1 - dr, R2 = L2 - dr, TF = R1 + G - 1;
    auto load = [&](const double* S, int seq, int L, int r, double (&v)[DP]) {
        for (int f = 0; f < DP; ++f) v[f] = (r >= 0 && r < L && f < d) ? S[(size_t(seq) * L + r) * d + f] : 0.0;
    };
    std::vector<WaveDm<C, DP, MODE>> dm(G);
    std::vector<WaveGy<C, DP, MODE>> gyl(G);
    std::vector<WaveFwd<C, LQ>> fw(G);
    std::vector<double> rowtot(size_t(R1 > 0 ? R1 : 1) * LQ, 0.0);
    double clev[LQ + 2];
    for (int p = 0; p < LQ + 2; ++p) clev[p] = (p >= 1 && p <= M) ? clev_in[p] : 0.0;
    const int last_lane = R2 > 0 ? (R2 - 1) / C : 0;
    for (int l = 0; l < G; ++l) {
        double ypts[C + 1][DP];
        for (int c = 0; c <= C; ++c) load(Y, j, L2, C * l + c, ypts[c]);
        int nv = R2 - C * l;
        nv = nv < 0 ? 0 : (nv > C ? C : nv);
        dm[l].set_y(ypts, nv);
        gyl[l].set_y(ypts, nv);
        fw[l].reset();
        if (MODE != MODE_PT_NODIFF) { double x0[DP]; load(X, i, L1, 0, x0); dm[l].prime(x0, kind, p0, p1); }
    }
    for (int t = 0; t < TF; ++t) {
        std::vector<std::array<double, LQ + 2>> snap(G);
        for (int l = 0; l < G; ++l)
            for (int m = 0; m < LQ + 2; ++m) snap[l][m] = l > 0 ? fw[l - 1].sout[m] : 0.0;
        for (int l = 0; l < G; ++l) {
            const int a = t - l;
            if (a < 0 || a >= R1) continue;
            double cin[LQ + 2], xn[DP], dmv[C];
            for (int m = 0; m < LQ + 2; ++m) cin[m] = snap[l][m];
            cin[0] = 0.0;
            load(X, i, L1, a + dr, xn);
            dm[l].row(xn, true, kind, p0, p1, dmv);
            fw[l].step(dmv, cin, M);
            if (l == last_lane)
                for (int m = 1; m <= LQ; ++m) rowtot[size_t(a) * LQ + m - 1] = m < M ? fw[l].sout[m] : 0.0;
        }
    }
    std::vector<WaveUndo<C, LQ>> bw(G);
    for (int l = 0; l < G; ++l) {
        bw[l].init(fw[l]);
        double xl[DP];
        load(X, i, L1, R1, xl);            // difference modes: x_{R1}; no-difference: row R1 does not exist (zeros), unused
        gyl[l].prime(xl, kind, p0, p1);
    }
    for (int u = 0; u < TF; ++u) {
        std::vector<std::array<double, 2 * LQ>> snap(G);
        for (int l = 0; l < G; ++l)
            for (int p = 0; p < LQ; ++p) {
                snap[l][p] = l < G - 1 ? bw[l + 1].sufout[p] : 0.0;
                snap[l][LQ + p] = l < G - 1 ? bw[l + 1].svout[p] : 0.0;
            }
        for (int l = 0; l < G; ++l) {
            const int a = R1 - 1 - (u - (G - 1 - l));
            if (a < 0 || a >= R1) continue;
            double sufin[LQ], svin[LQ], rt[LQ], xn[DP], dmv[C], lv[C];
            for (int p = 0; p < LQ; ++p) { sufin[p] = snap[l][p]; svin[p] = snap[l][LQ + p]; rt[p] = rowtot[size_t(a) * LQ + p]; }
            load(X, i, L1, a, xn);
            gyl[l].row(xn, kind, p0, p1, dmv);
            bw[l].step(dmv, clev, rt, sufin, svin, M, a == 0, l == 0, lv);
            gyl[l].contract(lv);
            if (a == 0) gyl[l].finish_pair();
        }
    }
    double gp0 = 0.0;
    for (int l = 0; l < G; ++l) {
        gp0 += gyl[l].gp0;
        const int npts = MODE == MODE_PT_NODIFF ? gyl[l].nvalid : (gyl[l].nvalid > 0 ? gyl[l].nvalid + 1 : 0);
        for (int c = 0; c < npts; ++c) {
            const int q = C * l + c;
            for (int f = 0; f < d; ++f) {
                double v;
                if (MODE == MODE_INC) v = (c > 0 ? gyl[l].g[c - 1][f] : 0.0) - (c < gyl[l].nvalid ? gyl[l].g[c][f] : 0.0);
                else v = gyl[l].g[c][f];
                gy[size_t(q) * d + f] += v;
            }
        }
    }
    return gp0;
}

template <int G, int C, int DP, int LQ>
double wave2_pair_mode(int mode, const double* X, const double* Y, int i, int j, int L1, int L2, int d, int M, int kind, double p0, double p1,
                       const double* clev, double* gy) {
    if (mode == MODE_INC) return wave2_pair<G, C, DP, LQ, MODE_INC>(X, Y, i, j, L1, L2, d, M, kind, p0, p1, clev, gy);
    if (mode == MODE_PT_DIFF) return wave2_pair<G, C, DP, LQ, MODE_PT_DIFF>(X, Y, i, j, L1, L2, d, M, kind, p0, p1, clev, gy);
    return wave2_pair<G, C, DP, LQ, MODE_PT_NODIFF>(X, Y, i, j, L1, L2, d, M, kind, p0, p1, clev, gy);
}
}  // namespace

extern "C" {
// Scratch-free wave formulation: every pair is swept twice, once per role, each sweep yielding the gradient of its
// register-resident side.  Same contract as emu_seq_grad.
int emu_seq_grad_wave2(const double* X, const double* Y, int N1, int N2, int L1, int L2, int d, int M, int kind, int mode, double p0, double p1,
                       int diag, const double* G, double* gX, double* gY, double* gbase, int Gg, int Cc) {
    const int DP = pad_of(d);
    const bool sym = !diag && !Y;
    if (diag || sym) { N2 = N1; L2 = L1; Y = X; gY = gX; }
    const int dr = mode == MODE_PT_NODIFF ? 0 : 1;
    if (DP != 4 && DP != 8) return -2;
    if (L1 - dr > Gg * Cc || L2 - dr > Gg * Cc || M > 8) return -2;
    std::fill(gX, gX + size_t(N1) * L1 * d, 0.0);
    if (gY != gX) std::fill(gY, gY + size_t(N2) * L2 * d, 0.0);
    std::vector<double> clev(M + 1);
    double gp0 = 0.0;
    for (int i = 0; i < N1; ++i)
        for (int j = diag ? i : 0; j < (diag ? i + 1 : N2); ++j) {
            for (int m = 0; m <= M; ++m) clev[m] = diag ? G[size_t(m) * N1 + i] : G[(size_t(m) * N1 + i) * N2 + j];
#define WP2(GG, CC, DD, LL, XX, YY, II, JJ, LA, LB, OUT) wave2_pair_mode<GG, CC, DD, LL>(mode, XX, YY, II, JJ, LA, LB, d, M, kind, p0, p1, clev.data(), OUT)
#define BOTH(GG, CC, DD, LL)                                                                  \
    do {                                                                                      \
        gp0 += WP2(GG, CC, DD, LL, X, Y, i, j, L1, L2, gY + size_t(j) * L2 * d);              \
        WP2(GG, CC, DD, LL, Y, X, j, i, L2, L1, gX + size_t(i) * L1 * d);                     \
    } while (0)
            const bool small = M <= 5;
            if (Gg == 16 && Cc == 2) { if (DP == 4) { if (small) BOTH(16, 2, 4, 4); else BOTH(16, 2, 4, 7); } else { if (small) BOTH(16, 2, 8, 4); else BOTH(16, 2, 8, 7); } }
            else if (Gg == 16 && Cc == 4) { if (DP == 4) { if (small) BOTH(16, 4, 4, 4); else BOTH(16, 4, 4, 7); } else { if (small) BOTH(16, 4, 8, 4); else BOTH(16, 4, 8, 7); } }
            else if (Gg == 64 && Cc == 2) { if (DP == 4) { if (small) BOTH(64, 2, 4, 4); else BOTH(64, 2, 4, 7); } else { if (small) BOTH(64, 2, 8, 4); else BOTH(64, 2, 8, 7); } }
            else return -2;
#undef BOTH
#undef WP2
        }
    if (gbase) { gbase[0] = gp0; gbase[1] = 0.0; }
    return 0;
}
}
