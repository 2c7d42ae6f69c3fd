/* sigkern_ref.c -- C restatement of the level recursions of the reference (TEST INFRASTRUCTURE, like the rest of oracle/:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; gpsig_amd never does).
 *
 * What it restates, operation for operation but one PAIR at a time so that a pair's lattice stays in cache (the reference
 * applies each op to the whole (N1, L1, N2, L2) tensor, which does not fit any memory at the benchmark sizes):
 *   gpsig/kernels.py:225-230          M = kappa(x_a, y_b): inner products (linear, :799-806) or exp(-|x - y|^2 / 2) (RBF, :862-864
 *                                     with _square_dist :765-776: |x|^2 + |y|^2 - 2 <x, y>)
 *   gpsig/signature_algs.py:25-26     double difference of M
 *   gpsig/signature_algs.py:28        K_1 = sum M
 *   gpsig/signature_algs.py:31-33     R = M * excumsum(excumsum(R, axis a), axis b);  K_m = sum R
 *   gpsig/kernels.py:322-333, gpsig/signature_algs.py:114-125   the same for inducing tensors vs sequences (chains along time)
 * It exists to time "the reference's CPU path" on all host cores (OpenMP over pairs) next to the NumPy oracle, SURVEY.md 8(d)
 * baseline (ii); tests/test_oracle.py pins it to the NumPy oracle.  Compiled by __graft_entry__.build() /
 * oracle/cref.py with gcc -O3 -fopenmp into oracle/_build/libsigkern_ref.so. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double kappa(const double* x, const double* y, int d, int base) {
    double in = 0.0, xs = 0.0, ys = 0.0;
    for (int f = 0; f < d; ++f) { in += x[f] * y[f]; xs += x[f] * x[f]; ys += y[f] * y[f]; }
    if (base == 0) return in;
    return exp(-(xs + ys - 2.0 * in) / 2.0);
}

/* X (n1, L1, d), Y (n2, L2, d) scaled sequences -> out (M+1, n1, n2): SignatureKernel._K_seq, first-order algorithm */
void sigkern_seq_levels(const double* X, const double* Y, int n1, int n2, int L1, int L2, int d, int M, int base, int difference,
                        double* out) {
    const int R1 = difference ? L1 - 1 : L1, R2 = difference ? L2 - 1 : L2;
#pragma omp parallel
    {
        double* K = (double*)malloc(sizeof(double) * (size_t)L1 * L2);
        double* dM = (double*)malloc(sizeof(double) * (size_t)(R1 > 0 ? R1 : 1) * (R2 > 0 ? R2 : 1));
        double* R = (double*)malloc(sizeof(double) * (size_t)(R1 > 0 ? R1 : 1) * (R2 > 0 ? R2 : 1));
        double* S = (double*)malloc(sizeof(double) * (size_t)(R1 > 0 ? R1 : 1) * (R2 > 0 ? R2 : 1));
#pragma omp for collapse(2) schedule(static)
        for (int i = 0; i < n1; ++i)
            for (int j = 0; j < n2; ++j) {
                const double *x = X + (size_t)i * L1 * d, *y = Y + (size_t)j * L2 * d;
                for (int a = 0; a < L1; ++a)
                    for (int b = 0; b < L2; ++b) K[a * L2 + b] = kappa(x + a * d, y + b * d, d, base);               /* kernels.py:225-230 */
                if (difference) {
                    for (int a = 0; a < R1; ++a)
                        for (int b = 0; b < R2; ++b)                                                                 /* signature_algs.py:26 */
                            dM[a * R2 + b] = K[(a + 1) * L2 + b + 1] + K[a * L2 + b] - K[a * L2 + b + 1] - K[(a + 1) * L2 + b];
                } else {
                    memcpy(dM, K, sizeof(double) * (size_t)L1 * L2);
                }
                out[(size_t)0 * n1 * n2 + (size_t)i * n2 + j] = 1.0;                                                 /* :19-23 */
                double sum = 0.0;
                for (int c = 0; c < R1 * R2; ++c) { R[c] = dM[c]; sum += dM[c]; }
                if (M >= 1) out[(size_t)1 * n1 * n2 + (size_t)i * n2 + j] = sum;                                     /* :28 */
                for (int m = 2; m <= M; ++m) {                                                                       /* :31 */
                    for (int b = 0; b < R2; ++b) {                                                                   /* excumsum along a */
                        double run = 0.0;
                        for (int a = 0; a < R1; ++a) { S[a * R2 + b] = run; run += R[a * R2 + b]; }
                    }
                    sum = 0.0;
                    for (int a = 0; a < R1; ++a) {                                                                   /* excumsum along b, * M, sum */
                        double run = 0.0;
                        for (int b = 0; b < R2; ++b) {
                            const double v = dM[a * R2 + b] * run;                                                   /* :32 */
                            run += S[a * R2 + b];
                            R[a * R2 + b] = v;
                            sum += v;
                        }
                    }
                    out[(size_t)m * n1 * n2 + (size_t)i * n2 + j] = sum;                                             /* :33 */
                }
            }
        free(K); free(dM); free(R); free(S);
    }
}

/* Z (lt, T, E, d) scaled tensor components (E = 2: increments, kernels.py:328-330), X (n, L, d) -> out (M+1, T, n):
 * SignatureKernel._K_tens_vs_seq + signature_kern_tens_vs_seq_first_order */
void sigkern_tens_vs_seq_levels(const double* Z, const double* X, int T, int n, int L, int d, int M, int E, int base, int difference,
                                double* out) {
    const int lt = M * (M + 1) / 2, R = difference ? L - 1 : L;
#pragma omp parallel
    {
        double* Mk = (double*)malloc(sizeof(double) * (size_t)lt * (R > 0 ? R : 1));
        double* Rv = (double*)malloc(sizeof(double) * (size_t)(R > 0 ? R : 1));
#pragma omp for collapse(2) schedule(static)
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < n; ++i) {
                const double* x = X + (size_t)i * L * d;
                for (int k = 0; k < lt; ++k) {
                    double prev = 0.0;
                    for (int tau = 0; tau < L; ++tau) {
                        const double* z = Z + (((size_t)k * T + t) * E) * d;
                        double v = E == 2 ? kappa(z + d, x + tau * d, d, base) - kappa(z, x + tau * d, d, base)       /* kernels.py:328-330 */
                                          : kappa(z, x + tau * d, d, base);                                          /* :332-333 */
                        if (difference) { if (tau > 0) Mk[k * R + tau - 1] = v - prev; prev = v; }                   /* signature_algs.py:114 */
                        else Mk[k * R + tau] = v;
                    }
                }
                out[(size_t)0 * T * n + (size_t)t * n + i] = 1.0;                                                    /* :116 */
                int k = 0;
                for (int lev = 1; lev <= M; ++lev) {                                                                 /* :119 */
                    for (int tau = 0; tau < R; ++tau) Rv[tau] = Mk[k * R + tau];                                     /* :120 */
                    ++k;
                    for (int j = 1; j < lev; ++j) {                                                                  /* :122 */
                        double run = 0.0;
                        for (int tau = 0; tau < R; ++tau) { const double r = Rv[tau]; Rv[tau] = Mk[k * R + tau] * run; run += r; }   /* :123 */
                        ++k;
                    }
                    double sum = 0.0;
                    for (int tau = 0; tau < R; ++tau) sum += Rv[tau];
                    out[(size_t)lev * T * n + (size_t)t * n + i] = sum;                                              /* :125 */
                }
            }
        free(Mk); free(Rv);
    }
}

int sigkern_ref_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
