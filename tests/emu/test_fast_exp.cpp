// Host check of gpsig_amd/csrc/fast_exp.hpp (the same arithmetic the device runs: fma, rint, ldexp are exact operations)
// against the long-double library exp.  Prints the worst error in ulps for both entry points.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "fast_exp.hpp"

static const double tab[64] = {GPSIG_EXP2_TABLE};
static const double tab256[256] = {GPSIG_EXP2_TABLE256};
static const double tab1024[1024] = {GPSIG_EXP2_TABLE1024};
static const double tab2048[2048] = {GPSIG_EXP2_TABLE2048};

static double ulps(double got, long double want) {
    if (want == 0.0L) return got == 0.0 ? 0.0 : 1e9;
    int e;
    frexpl(want, &e);
    const long double ulp = ldexpl(1.0L, e - 53);
    return double(fabsl((long double)got - want) / ulp);
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    unsigned long long s = 88172645463325252ull;
    double worst1 = 0, worst2 = 0, worst1024 = 0, worst2048 = 0, worst2l = 0;
    double tab2l[64];                                                     // what exp_tab2l_fill puts into LDS
    for (int j = 0; j < 64; ++j) tab2l[j] = j < 32 ? tab1024[32 * j] : tab1024[j - 32];
    for (long i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = double(s >> 11) / 9007199254740992.0;
        double a;
        switch (i & 3) {
            case 0: a = -700.0 * u; break;
            case 1: a = -40.0 * u; break;
            case 2: a = -1.0 * u * u; break;
            default: a = 700.0 * (u - 0.5); break;
        }
        const double g1 = gpsig::kexp_tab(a, tab);
        const double w1 = ulps(g1, expl((long double)a));
        if (w1 > worst1) worst1 = w1;
        const double t = a * gpsig::EXP_T_PER_A;                          // the argument kexp2_tab sees IS t: reference 2^(t/64)
        const double g2 = gpsig::kexp2_tab(t, tab);
        const double w2 = ulps(g2, exp2l((long double)t / 64.0L));
        if (w2 > worst2) worst2 = w2;
        const double g3 = gpsig::kexp2_tab256(4.0 * t, tab256);                 // the 256-entry variant: 2^(t'/256), t' = 4 t
        const double w3 = ulps(g3, exp2l((long double)(4.0 * t) / 256.0L));
        if (w3 > worst2) worst2 = w3;
        const double w4 = ulps(gpsig::kexp2_tabn<1024>(16.0 * t, tab1024), exp2l((long double)(16.0 * t) / 1024.0L));   // 2^(t'/1024)
        if (w4 > worst1024) worst1024 = w4;
        const double w5 = ulps(gpsig::kexp2_tabn<2048>(32.0 * t, tab2048), exp2l((long double)(32.0 * t) / 2048.0L));
        if (w5 > worst2048) worst2048 = w5;
        const double w6 = ulps(gpsig::kexp2_tab2l(16.0 * t, tab2l), exp2l((long double)(16.0 * t) / 1024.0L));          // two tables of 32 entries
        if (w6 > worst2l) worst2l = w6;
    }
    // edge cases: huge negative arguments give 0, zero gives 1
    const double e0 = gpsig::kexp_tab(0.0, tab), e1 = gpsig::kexp_tab(-1e300, tab), e2 = gpsig::kexp2_tab(-1e300, tab),
                 e3 = gpsig::kexp_tab(-800.0, tab), e4 = gpsig::kexp2_tab(0.0, tab);
    const bool e256 = gpsig::kexp2_tab256(0.0, tab256) == 1.0 && gpsig::kexp2_tab256(-1e300, tab256) == 0.0 &&
                      fabs(gpsig::EXP_PRESCALE256 * gpsig::EXP_PRESCALE256 / (4.0 * gpsig::EXP_T_PER_A) - 1.0) < 4e-16;
    const bool en = gpsig::kexp2_tabn<1024>(0.0, tab1024) == 1.0 && gpsig::kexp2_tabn<1024>(-1e300, tab1024) == 0.0 &&
                    gpsig::kexp2_tabn<2048>(0.0, tab2048) == 1.0 && gpsig::kexp2_tabn<2048>(-1e300, tab2048) == 0.0 &&
                    fabs(gpsig::ExpTabN<1024>::PRESCALE * gpsig::ExpTabN<1024>::PRESCALE / (16.0 * gpsig::EXP_T_PER_A) - 1.0) < 4e-16 &&
                    fabs(gpsig::ExpTabN<2048>::PRESCALE * gpsig::ExpTabN<2048>::PRESCALE / (32.0 * gpsig::EXP_T_PER_A) - 1.0) < 4e-16;
    const bool e2l = gpsig::kexp2_tab2l(0.0, tab2l) == 1.0 && gpsig::kexp2_tab2l(-1e300, tab2l) == 0.0;
    const bool scale_ok = e256 && en && e2l && fabs(gpsig::EXP_PRESCALE * gpsig::EXP_PRESCALE / gpsig::EXP_T_PER_A - 1.0) < 4e-16;
    printf("%.4f %.4f %.4f %.4f %.4f %d\n", worst1, worst2, worst1024, worst2048, worst2l, int(e0 == 1.0 && e1 == 0.0 && e2 == 0.0 && e3 == 0.0 && e4 == 1.0 && scale_ok));
    return 0;
}
