// seq_configs.hpp -- which compile-time shapes of the seq-gram kernel exist, how one is chosen for a
// problem, and the record geometry that goes with it.  HIP-free (the CPU emulator uses the same code).
#pragma once

#include <stdint.h>

#include "seq_core.hpp"

namespace gpsig {

// G: lanes per pair (16 = one DPP row, 4 pairs per wave; 64 = whole wave, 1 pair per wave)
// C: lattice columns per lane;  D: padded state-space dimension;  MMAX: levels (exact or upper bound)
struct SeqConfig {
    int G, C, D, MMAX;
    bool exact;
};

// The product build instantiates every entry for MODE_INC and MODE_PT_DIFF; entries with exact == false
// also for MODE_PT_NODIFF.  X(G, C, D, MMAX, EXACT)
#define GPSIG_SEQ_CONFIGS_EXACT(X) \
    X(16, 2, 4, 4, true)  X(16, 2, 4, 5, true)  \
    X(16, 4, 4, 4, true)  X(16, 4, 4, 5, true)  \
    X(16, 4, 8, 4, true)  X(16, 4, 8, 5, true)  \
    X(64, 2, 16, 6, true)
// num_levels 4 and 5 fixed at compile time for the other shapes up to D = 16 (float64): the run-time-M kernels of these shapes
// sit at one wavefront per SIMD, where the level predicates cost a factor 1.7-2 (C5 in float64: 301 -> 179 ms)
#define GPSIG_SEQ_CONFIGS_EX_G16_D4(X) X(16, 1, 4, 4, true) X(16, 1, 4, 5, true) X(16, 8, 4, 4, true) X(16, 8, 4, 5, true)
#define GPSIG_SEQ_CONFIGS_EX_G16_D8(X) X(16, 1, 8, 4, true) X(16, 1, 8, 5, true) X(16, 2, 8, 4, true) X(16, 2, 8, 5, true) X(16, 8, 8, 4, true) X(16, 8, 8, 5, true)
#define GPSIG_SEQ_CONFIGS_EX_G16_D16(X) X(16, 1, 16, 4, true) X(16, 1, 16, 5, true) X(16, 2, 16, 4, true) X(16, 2, 16, 5, true) X(16, 4, 16, 4, true) X(16, 4, 16, 5, true)
#define GPSIG_SEQ_CONFIGS_EX_G64_D4(X) X(64, 1, 4, 4, true) X(64, 1, 4, 5, true) X(64, 2, 4, 4, true) X(64, 2, 4, 5, true) X(64, 4, 4, 4, true) X(64, 4, 4, 5, true) X(64, 8, 4, 4, true) X(64, 8, 4, 5, true)
#define GPSIG_SEQ_CONFIGS_EX_G64_D8(X) X(64, 1, 8, 4, true) X(64, 1, 8, 5, true) X(64, 2, 8, 4, true) X(64, 2, 8, 5, true) X(64, 4, 8, 4, true) X(64, 4, 8, 5, true) X(64, 8, 8, 4, true) X(64, 8, 8, 5, true)
#define GPSIG_SEQ_CONFIGS_EX_G64_D16(X) X(64, 1, 16, 4, true) X(64, 1, 16, 5, true) X(64, 2, 16, 4, true) X(64, 2, 16, 5, true) X(64, 4, 16, 4, true) X(64, 4, 16, 5, true)
#define GPSIG_SEQ_CONFIGS_EX_MORE(X) GPSIG_SEQ_CONFIGS_EX_G16_D4(X) GPSIG_SEQ_CONFIGS_EX_G16_D8(X) GPSIG_SEQ_CONFIGS_EX_G16_D16(X) GPSIG_SEQ_CONFIGS_EX_G64_D4(X) GPSIG_SEQ_CONFIGS_EX_G64_D8(X) GPSIG_SEQ_CONFIGS_EX_G64_D16(X)
#define GPSIG_SEQ_CONFIGS_G16(X) \
    X(16, 1, 4, 8, false) X(16, 2, 4, 8, false) X(16, 4, 4, 8, false) X(16, 8, 4, 8, false) \
    X(16, 1, 8, 8, false) X(16, 2, 8, 8, false) X(16, 4, 8, 8, false) X(16, 8, 8, 8, false) \
    X(16, 1, 16, 8, false) X(16, 2, 16, 8, false) X(16, 4, 16, 8, false) \
    X(16, 1, 32, 8, false) X(16, 2, 32, 8, false)
#define GPSIG_SEQ_CONFIGS_G64(X) \
    X(64, 1, 4, 8, false) X(64, 2, 4, 8, false) X(64, 4, 4, 8, false) X(64, 8, 4, 8, false) \
    X(64, 1, 8, 8, false) X(64, 2, 8, 8, false) X(64, 4, 8, 8, false) X(64, 8, 8, 8, false) \
    X(64, 1, 16, 8, false) X(64, 2, 16, 8, false) X(64, 4, 16, 8, false) \
    X(64, 1, 32, 8, false) X(64, 2, 32, 8, false)
#define GPSIG_SEQ_CONFIGS_GENERIC(X) GPSIG_SEQ_CONFIGS_G16(X) GPSIG_SEQ_CONFIGS_G64(X)
// SignatureSpectral (float64, MODE_PT_DIFF, first order; compile-time family: seq_step_spectral): run-time num_levels, d <= 16
#define GPSIG_SEQ_CONFIGS_SPECTRAL_G16(X) \
    X(16, 1, 4, 8, false) X(16, 2, 4, 8, false) X(16, 4, 4, 8, false) X(16, 8, 4, 8, false) \
    X(16, 1, 8, 8, false) X(16, 2, 8, 8, false) X(16, 4, 8, 8, false) X(16, 8, 8, 8, false) \
    X(16, 1, 16, 8, false) X(16, 2, 16, 8, false) X(16, 4, 16, 8, false)
#define GPSIG_SEQ_CONFIGS_SPECTRAL_G64(X) \
    X(64, 1, 4, 8, false) X(64, 2, 4, 8, false) X(64, 4, 4, 8, false) X(64, 8, 4, 8, false) \
    X(64, 1, 8, 8, false) X(64, 2, 8, 8, false) X(64, 4, 8, 8, false) X(64, 8, 8, 8, false) \
    X(64, 1, 16, 8, false) X(64, 2, 16, 8, false) X(64, 4, 16, 8, false)
#define GPSIG_SEQ_CONFIGS_SPECTRAL(X) GPSIG_SEQ_CONFIGS_SPECTRAL_G16(X) GPSIG_SEQ_CONFIGS_SPECTRAL_G64(X)
// float32 halves the register cost of a lane's columns: the widest shapes exist for it only; the exact one is the
// shape of BASELINE.json configs[4] (L=128, d=16, num_levels=6)
#define GPSIG_SEQ_CONFIGS_F32_EXACT(X) X(16, 8, 16, 6, true) X(16, 4, 8, 5, true) \
    X(16, 2, 4, 4, true) X(16, 2, 4, 5, true) X(16, 4, 4, 4, true) X(16, 4, 4, 5, true) X(16, 4, 8, 4, true)
#define GPSIG_SEQ_CONFIGS_F32_G16(X) GPSIG_SEQ_CONFIGS_G16(X) X(16, 8, 16, 8, false)
#define GPSIG_SEQ_CONFIGS_F32_G64(X) GPSIG_SEQ_CONFIGS_G64(X) X(64, 8, 16, 8, false)
// float32: num_levels 4 / 5 at compile time for the same shapes as float64 (a factor 1.4 on the run-time-M kernels)
#define GPSIG_SEQ_CONFIGS_F32_ALL(X) GPSIG_SEQ_CONFIGS_F32_EXACT(X) GPSIG_SEQ_CONFIGS_EX_MORE(X) GPSIG_SEQ_CONFIGS_F32_G16(X) GPSIG_SEQ_CONFIGS_F32_G64(X)
#define GPSIG_SEQ_CONFIGS_ALL(X) GPSIG_SEQ_CONFIGS_EXACT(X) GPSIG_SEQ_CONFIGS_EX_MORE(X) GPSIG_SEQ_CONFIGS_GENERIC(X)

// Higher-order kernels (run-time num_levels <= MMAX, run-time order <= OMAX), MODE_INC and MODE_PT_DIFF.
// X(G, C, D, MMAX, OMAX).  State and temporaries grow with C * OMAX^2, so wide lanes come with small orders.
#define GPSIG_SEQ_HO_SHAPES(X, D_) \
    X(16, 4, D_, 6, 2) X(16, 2, D_, 6, 4) X(64, 1, D_, 8, 8) X(64, 2, D_, 6, 4) X(64, 4, D_, 6, 2) X(64, 8, D_, 6, 2)
#define GPSIG_SEQ_HO_D4(X) GPSIG_SEQ_HO_SHAPES(X, 4)
#define GPSIG_SEQ_HO_D8(X) GPSIG_SEQ_HO_SHAPES(X, 8)
#define GPSIG_SEQ_HO_D16(X) GPSIG_SEQ_HO_SHAPES(X, 16)
// 32-wide state spaces: only the narrow lanes fit the register file
#define GPSIG_SEQ_HO_D32(X) X(16, 2, 32, 6, 2) X(64, 1, 32, 8, 8) X(64, 2, 32, 6, 4)
#define GPSIG_SEQ_HO_ALL(X) GPSIG_SEQ_HO_D4(X) GPSIG_SEQ_HO_D8(X) GPSIG_SEQ_HO_D16(X) GPSIG_SEQ_HO_D32(X)

struct SeqHOConfig {
    int G, C, D, MMAX, OMAX;
};
// cheapest higher-order shape that fits (rows, d, num_levels, order); -1 if none
inline int seq_select_ho(const SeqHOConfig* tab, int n, int Ry, int d, int M, int order) {
    int best = -1;
    long best_cost = 0;
    for (int k = 0; k < n; ++k) {
        const SeqHOConfig& c = tab[k];
        if (c.MMAX < M || c.OMAX < order || c.D < d || long(c.G) * c.C < Ry) continue;
        const long cost = (long(c.G) * c.C * (c.D + 4L * c.OMAX * c.OMAX)) * 2 + (c.G == 16 ? 0 : 1);
        if (best < 0 || cost < best_cost) { best = k; best_cost = cost; }
    }
    return best;
}

// Pick the cheapest config that fits: y-side record rows Ry <= G*C, d <= D, levels M (== MMAX if exact,
// <= MMAX otherwise).  Cost = lanes*columns*D actually paid per pair (G*C*D); ties go to the exact
// variant, then to the smaller group (more pairs per wave).  Returns -1 if nothing fits.
inline int seq_select(const SeqConfig* tab, int n, int Ry, int d, int M, bool allow_exact) {
    int best = -1;
    long best_cost = 0;
    for (int k = 0; k < n; ++k) {
        const SeqConfig& c = tab[k];
        if (c.exact && (!allow_exact || c.MMAX != M)) continue;
        if (!c.exact && c.MMAX < M) continue;
        if (c.D < d || long(c.G) * c.C < Ry) continue;
        const long cost = (long(c.G) * c.C * c.D) * 4 + (c.exact ? 0 : 2) + (c.G == 16 ? 0 : 1);
        if (best < 0 || cost < best_cost) { best = k; best_cost = cost; }
    }
    return best;
}

// Record geometry.  A "record" is what the kernel streams for one sequence: `rows` rows of D values,
// RS = D + (16 bytes worth of padding) apart so that the 16 lanes of a pair group, which read 16
// consecutive rows with 16-byte LDS loads, hit 16 different bank groups; the whole record is padded to
// a multiple of 1 KiB so it moves as whole 64-lane x 16-byte pieces.
//   MODE_INC        rows = [0 ; the L-1 increments]          (difference)      -> L rows
//                   rows = [0 ; the L points]                (no difference)   -> L+1 rows
//   MODE_PT_DIFF    rows = the L points                                        -> L rows
//   MODE_PT_NODIFF  rows = [0 ; the L points]                                  -> L+1 rows
// The leading zero row is the lattice's "pair boundary" row on the x side and the dummy column of
// lane 0 on the y side.
struct SeqGeom {
    int mode, rows, RS, rec_elems;
};
inline SeqGeom seq_geometry(int base_kind, int difference, int L, int D, int elem_bytes) {
    SeqGeom g;
    g.mode = base_kind == BASE_LINEAR ? MODE_INC : (difference ? MODE_PT_DIFF : MODE_PT_NODIFF);
    g.rows = (g.mode == MODE_INC) ? (difference ? L : L + 1) : (g.mode == MODE_PT_DIFF ? L : L + 1);
    const int vec = 16 / elem_bytes;
    g.RS = D + vec;
    const int piece = 64 * vec;
    g.rec_elems = (g.rows * g.RS + piece - 1) / piece * piece;
    return g;
}

}  // namespace gpsig
