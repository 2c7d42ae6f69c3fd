// fast_exp.hpp -- float64 exp for the base kernels' envelopes (RBF: exp(-|x-y|^2/2), gpsig/kernels.py:862-864; Matern:
// exp(-c r), :955-993) at about 13 vector instructions instead of the ~20 of the library routine.
//
//     exp(a) = 2^(t/64),  t = a * 64/ln2 = n + r,  n = rint(t) = 64 k + j,  |r| <= 1/2
//            = 2^k * 2^(j/64) * exp(r ln2/64)
// 2^(j/64) comes from a 64-entry table (512 B: each entry on its own pair of LDS banks, so a wavefront's 64 different
// indices are read without a bank conflict beyond the two passes a 64-lane b64 read takes anyway), exp(r ln2/64) - 1 from a
// degree-5 Taylor polynomial (|r ln2/64| <= 0.0055: the first omitted term is 3.5e-17 relative), 2^k by v_ldexp_f64 (which
// also flushes to zero where the result underflows; v_cvt_i32_f64 saturates, so any finite argument is safe).
// Measured against the long-double exp over [-745, 350]: <= 1.3 ulp (half an ulp of it is the rounding of the table entry;
// tests/emu/test_fast_exp.cpp, run by tests/test_device_headers.py).
//
// Two entries: kexp_tab(a, tab) takes the argument itself; kexp2_tab(t, tab) takes t = a * 64/ln2 already scaled -- a
// kernel that forms a = <x,z> - |x|^2/2 - |z|^2/2 from points it prepared itself scales the points by sqrt(64/ln2) once
// (EXP_PRESCALE) and saves the multiplication and the two-step reduction.
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GPSIG_FX __host__ __device__ __forceinline__
#else
#define GPSIG_FX inline
#endif

namespace gpsig {

constexpr int EXP_TAB_N = 64;
constexpr double EXP_T_PER_A = 0x1.71547652b82fep+6;        // 64 / ln 2
constexpr double EXP_PRESCALE = 0x1.337cc2183b050p+3;         // sqrt(64 / ln 2): points scaled by this give t instead of a

// 2^(j/64), j = 0..63, correctly rounded
#define GPSIG_EXP2_TABLE                                                                                    \
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,                 \
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,                 \
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,                 \
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,                 \
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,                 \
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,                 \
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,                 \
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,                 \
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,                 \
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,                 \
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,                 \
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,                 \
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,                 \
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,                 \
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,                 \
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0

#if defined(__HIPCC__)
static __device__ const double g_exp2_tab[EXP_TAB_N] = {GPSIG_EXP2_TABLE};

// Every thread of the workgroup calls this once (before a barrier / before the wave's first use for one-wave groups).
__device__ __forceinline__ void exp_tab_fill(double* lds_tab, int tid, int nthreads) {
    for (int j = tid; j < EXP_TAB_N; j += nthreads) lds_tab[j] = g_exp2_tab[j];
}
#endif

// int(n), saturating: v_cvt_i32_f64 saturates by itself on the device
GPSIG_FX int exp_tab_int(double n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return int(n);
#else
    return n < -2147483648.0 ? int(-2147483647 - 1) : (n > 2147483647.0 ? 2147483647 : int(n));
#endif
}

// exp(r * ln2/64) - 1 = r * q(r), |r| <= 1/2; coefficients (ln2/64)^k / k!
GPSIG_FX double exp_tab_tail(double r) {
    double q = 0x1.5d87fe78a6731p-40;
    q = fma(q, r, 0x1.3b2ab6fba4e77p-31);
    q = fma(q, r, 0x1.c6b08d704a0c0p-23);
    q = fma(q, r, 0x1.ebfbdff82c58fp-15);
    q = fma(q, r, 0x1.62e42fefa39efp-7);
    return q * r;
}

// 2^(t/64) for any finite t (t = 64 a / ln 2)
GPSIG_FX double kexp2_tab(double t, const double* tab) {
    const double n = rint(t);                        // v_rndne_f64
    const double s = exp_tab_tail(t - n);            // t - n is exact
    const int ni = exp_tab_int(n);
    const double tj = tab[ni & (EXP_TAB_N - 1)];
    return ldexp(fma(tj, s, tj), ni >> 6);
}

// exp(a) for any a < 700 (arguments below -746 are clamped there: the result is 0 either way)
GPSIG_FX double kexp_tab(double a, const double* tab) {
    a = fmax(a, -746.0);
    const double n = rint(a * EXP_T_PER_A);
    double r = fma(n, -0x1.62e42fee00000p-7, a);     // a - n ln2/64 in two steps: the first product is exact for |n| < 2^21
    r = fma(n, -0x1.a39ef35793c76p-39, r);
    // polynomial in r itself (|r| <= ln2/128)
    double q = 1.0 / 120;
    q = fma(q, r, 1.0 / 24);
    q = fma(q, r, 1.0 / 6);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    const double s = q * r;
    const int ni = exp_tab_int(n);
    const double tj = tab[ni & (EXP_TAB_N - 1)];
    return ldexp(fma(tj, s, tj), ni >> 6);
}

}  // namespace gpsig
