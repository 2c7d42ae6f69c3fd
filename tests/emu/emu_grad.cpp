// emu_grad.cpp -- CPU harness around gpsig_amd/csrc/grad_core.hpp (TEST INFRASTRUCTURE).
// Runs the very per-pair code the gfx950 gradient kernels run, one pair after the other, with plain adds in place
// of atomics, so the adjoint recursions can be checked against torch.autograd on a machine without a GPU.
#include <cstdint>
#include <cstring>
#include <vector>

#include "grad_core.hpp"

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
                 const double* G, double* gZ, double* gX, double* levels_out, double* gbase) {
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
    switch (DP) {
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
int emu_tens_grad(const double* Z, int T, int d, int M, int kind, int incr, double p0, double p1, const double* G, double* gZ, double* gbase) {
    const int DP = pad_of(d), lt = M * (M + 1) / 2, E = incr ? 2 : 1;
    const int64_t rows = int64_t(lt) * T * E;
    std::vector<double> zp = pad_rows(Z, rows, d, DP), gzp(zp.size(), 0.0);
    double gb[2] = {0, 0};
    TensGradArgs A;
    memset(&A, 0, sizeof(A));
    A.z = zp.data(); A.gz = gzp.data(); A.T = T; A.M = M; A.kind = kind; A.incr = incr; A.p0 = p0; A.p1 = p1;
    A.G = G; A.gm = int64_t(T) * T; A.gt = T; A.gn = 1;
    A.gbase = gb;
    switch (DP) {
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
