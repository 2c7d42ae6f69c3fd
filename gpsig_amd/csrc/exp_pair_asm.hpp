// exp_pair_asm.hpp -- the table-driven 2^(t/N) of fast_exp.hpp for TWO arguments at once, as one block of hand-scheduled gfx950 instructions
// (device code only).  Used by the Kzx tile kernel (tvs_tile_kernel.hpp) and by the reverse sweeps of the point kernels (grad_wave_core.hpp).
#pragma once

#include "fast_exp.hpp"

#if defined(__HIPCC__)
namespace gpsig {

// ---- The table-driven 2^(t/N) of fast_exp.hpp (kexp2_tabn / kexp2_tab256: same operations, same order, same bits) for TWO arguments at once, as one
// block of hand-scheduled instructions: both roundings and both table reads first, the polynomial tails while the reads are in flight, one wait.
// Left to the compiler each exp is one dependent chain that issues its read behind its tail and waits for it at once (and any attempt to steer it
// with sched_barrier spilled the tensors' components).  11 vector instructions per exp.  tab: LDS byte address of the table (a scalar).
// NEG: the arguments are -t0, -t1 (the Matern families hand in q = s r and want 2^(-q/N): the sign rides on the source modifiers).
#define KEXP2_ASM_HEAD(SGN, BITS)                                                                        \
    "v_rndne_f64 %[r0], " SGN "%[t0]\n\tv_rndne_f64 %[r1], " SGN "%[t1]\n\t"                         \
    "v_cvt_i32_f64 %[i0], %[r0]\n\tv_cvt_i32_f64 %[i1], %[r1]\n\t"                                    \
    "v_bfe_u32 %[a0], %[i0], 0, " BITS "\n\tv_bfe_u32 %[a1], %[i1], 0, " BITS "\n\t"                  \
    "v_lshl_add_u32 %[a0], %[a0], 3, %[tab]\n\tv_lshl_add_u32 %[a1], %[a1], 3, %[tab]\n\t"            \
    "ds_read_b64 %[e0], %[a0]\n\tds_read_b64 %[e1], %[a1]\n\t"                                        \
    "v_add_f64 %[r0], " SGN "%[t0], -%[r0]\n\tv_add_f64 %[r1], " SGN "%[t1], -%[r1]\n\t"
#define KEXP2_ASM_TAIL(BITS)                                                                             \
    "v_mul_f64 %[r0], %[q0], %[r0]\n\tv_mul_f64 %[r1], %[q1], %[r1]\n\t"                              \
    "v_ashrrev_i32 %[i0], " BITS ", %[i0]\n\tv_ashrrev_i32 %[i1], " BITS ", %[i1]\n\t"                \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                          \
    "v_fma_f64 %[e0], %[e0], %[r0], %[e0]\n\tv_fma_f64 %[e1], %[e1], %[r1], %[e1]\n\t"                \
    "v_ldexp_f64 %[e0], %[e0], %[i0]\n\tv_ldexp_f64 %[e1], %[e1], %[i1]"
#define KEXP2_ASM_DEG3                                                                                   \
    "v_fma_f64 %[q0], %[c3], %[r0], %[c2]\n\tv_fma_f64 %[q1], %[c3], %[r1], %[c2]\n\t"                \
    "v_fma_f64 %[q0], %[q0], %[r0], %[c1]\n\tv_fma_f64 %[q1], %[q1], %[r1], %[c1]\n\t"
#define KEXP2_ASM_DEG4                                                                                   \
    "v_fma_f64 %[q0], %[c4], %[r0], %[c3]\n\tv_fma_f64 %[q1], %[c4], %[r1], %[c3]\n\t"                \
    "v_fma_f64 %[q0], %[q0], %[r0], %[c2]\n\tv_fma_f64 %[q1], %[q1], %[r1], %[c2]\n\t"                \
    "v_fma_f64 %[q0], %[q0], %[r0], %[c1]\n\tv_fma_f64 %[q1], %[q1], %[r1], %[c1]\n\t"
#define KEXP2_ASM_OUTS [e0] "=&v"(e0), [e1] "=&v"(e1), [r0] "=&v"(r0), [r1] "=&v"(r1), [q0] "=&v"(q0), [q1] "=&v"(q1), [i0] "=&v"(i0), [i1] "=&v"(i1), \
                      [a0] "=&v"(a0), [a1] "=&v"(a1)
template <int N, bool NEG = false>
__device__ __forceinline__ void kexp2_pair_asm(double t0, double t1, unsigned tab, double& e0, double& e1) {
    static_assert(N == 256 || N == 1024 || N == 2048, "table sizes with an asm form");
    double r0, r1, q0, q1;
    int i0, i1, a0, a1;
    if constexpr (N == 256) {
        const double c4 = 0x1.3b2ab6fba4e77p-39, c3 = 0x1.c6b08d704a0c0p-29, c2 = 0x1.ebfbdff82c58fp-19, c1 = 0x1.62e42fefa39efp-9;
        if constexpr (NEG)
            asm volatile(KEXP2_ASM_HEAD("-", "8") KEXP2_ASM_DEG4 KEXP2_ASM_TAIL("8") : KEXP2_ASM_OUTS
                         : [t0] "v"(t0), [t1] "v"(t1), [tab] "s"(tab), [c4] "s"(c4), [c3] "v"(c3), [c2] "s"(c2), [c1] "s"(c1));
        else
            asm volatile(KEXP2_ASM_HEAD("", "8") KEXP2_ASM_DEG4 KEXP2_ASM_TAIL("8") : KEXP2_ASM_OUTS
                         : [t0] "v"(t0), [t1] "v"(t1), [tab] "s"(tab), [c4] "s"(c4), [c3] "v"(c3), [c2] "s"(c2), [c1] "s"(c1));
    } else {
        const double c3 = ExpTabN<N>::C3, c2 = ExpTabN<N>::C2, c1 = ExpTabN<N>::C1;
        if constexpr (N == 1024 && NEG)
            asm volatile(KEXP2_ASM_HEAD("-", "10") KEXP2_ASM_DEG3 KEXP2_ASM_TAIL("10") : KEXP2_ASM_OUTS
                         : [t0] "v"(t0), [t1] "v"(t1), [tab] "s"(tab), [c3] "s"(c3), [c2] "v"(c2), [c1] "s"(c1));
        else if constexpr (N == 1024)
            asm volatile(KEXP2_ASM_HEAD("", "10") KEXP2_ASM_DEG3 KEXP2_ASM_TAIL("10") : KEXP2_ASM_OUTS
                         : [t0] "v"(t0), [t1] "v"(t1), [tab] "s"(tab), [c3] "s"(c3), [c2] "v"(c2), [c1] "s"(c1));
        else if constexpr (NEG)
            asm volatile(KEXP2_ASM_HEAD("-", "11") KEXP2_ASM_DEG3 KEXP2_ASM_TAIL("11") : KEXP2_ASM_OUTS
                         : [t0] "v"(t0), [t1] "v"(t1), [tab] "s"(tab), [c3] "s"(c3), [c2] "v"(c2), [c1] "s"(c1));
        else
            asm volatile(KEXP2_ASM_HEAD("", "11") KEXP2_ASM_DEG3 KEXP2_ASM_TAIL("11") : KEXP2_ASM_OUTS
                         : [t0] "v"(t0), [t1] "v"(t1), [tab] "s"(tab), [c3] "s"(c3), [c2] "v"(c2), [c1] "s"(c1));
    }
}

}  // namespace gpsig
#endif
