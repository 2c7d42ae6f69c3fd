// grad_wave_kernel.hpp -- gfx950 kernels of the wavefront-parallel sequence-pair gradient.
//
//   seq_grad_wave_kernel   the two lattice sweeps of grad_wave_core.hpp: 64/G pairs per wavefront, one DPP shift per
//                          handed-over word and step (row_shr/row_shl for G = 16, wave_shr/wave_shl for G = 64), forward
//                          Q's through an HBM scratch slot per pair group, Lam[a][b] out.
//   lam_contract_kernel    Lam -> gradients of the observations: Gam = adjoint of the double increment
//                          (signature_algs.py:26), times the base kernel's derivatives (grad_core.hpp: base_eval_grad).
#pragma once

#include <hip/hip_runtime.h>

#include "grad_wave_core.hpp"
#include "seq_args.hpp"

namespace gpsig {

template <int G>
__device__ __forceinline__ double wave_from_left(double v) {      // lane l <- lane l-1, 0 into the first lane of each group
    int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (G == 16) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true);     // row_shr:1
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);     // wave_shr:1
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
        if constexpr (G == 32) {                                            // two groups per wavefront: lane 32 must not see lane 31
            const int m = (threadIdx.x & 31) == 0 ? 0 : -1;
            lo &= m; hi &= m;
        }
    }
    return __hiloint2double(hi, lo);
}
template <int G>
__device__ __forceinline__ double wave_from_right(double v) {     // lane l <- lane l+1, 0 into the last lane of each group
    int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (G == 16) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xf, 0xf, true);     // row_shl:1
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x101, 0xf, 0xf, true);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);     // wave_shl:1
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
        if constexpr (G == 32) {
            const int m = (threadIdx.x & 31) == 31 ? 0 : -1;
            lo &= m; hi &= m;
        }
    }
    return __hiloint2double(hi, lo);
}

template <int DP>
__device__ __forceinline__ void wave_load_point(const double* __restrict__ S, int64_t seq, int L, int d, int r, double (&v)[DP]) {
    const double* p = S + (seq * L + r) * d;
    const bool ok = r >= 0 && r < L;
#pragma unroll
    for (int f = 0; f < DP; ++f) v[f] = (ok && f < d) ? p[f] : 0.0;
}

// grid: ngroups / (64 / G) workgroups of one wavefront
template <int G, int C, int DP, int LQ, int MODE>
__global__ void __launch_bounds__(64) seq_grad_wave_kernel(const WaveGradArgs A) {
    constexpr int PW = 64 / G;
    const int lane = threadIdx.x, lam = lane % G;
    const int grp = blockIdx.x * PW + lane / G;
    const int dr = MODE == MODE_PT_NODIFF ? 0 : 1;
    const int R1 = A.L1 - dr, R2 = A.L2 - dr, M = A.M;
    const int TF = R1 + G - 1;
    double* scr = A.scratch + size_t(grp) * size_t(M - 1) * TF * G * C;
    auto slot = [&](int m, int tf, int l, int c) -> double& { return scr[((size_t(m) * TF + tf) * G + l) * C + c]; };
    const int64_t rounds = (A.npairs + A.ngroups - 1) / A.ngroups;

    WaveDm<C, DP, MODE> dmg;
    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t pp = rd * A.ngroups + grp;
        const bool have = pp < A.npairs;
        const int64_t pg = A.pair0 + (have ? pp : 0);
        const int64_t i = A.diag ? pg : pg / A.N2, j = A.diag ? pg : pg % A.N2;
        {
            double ypts[C + 1][DP];
#pragma unroll
            for (int c = 0; c <= C; ++c) wave_load_point<DP>(A.Y, j, A.L2, A.d, C * lam + c, ypts[c]);
            int nv = R2 - C * lam;
            dmg.set_y(ypts, nv < 0 ? 0 : (nv > C ? C : nv));
        }
        double clev[LQ + 2];
#pragma unroll
        for (int p = 0; p < LQ + 2; ++p) clev[p] = (have && p >= 1 && p <= M) ? A.G[p * A.gm + i * A.gi + j * A.gj] : 0.0;

        // ---- forward sweep
        {
            WaveFwd<C, LQ> fw;
            fw.reset();
            if (MODE != MODE_PT_NODIFF) {
                double x0[DP];
                wave_load_point<DP>(A.X, i, A.L1, A.d, 0, x0);
                dmg.prime(x0, A.kind, A.p0, A.p1);
            }
            double xcur[DP];                                  // the point row of the coming step, loaded one step ahead
            wave_load_point<DP>(A.X, i, A.L1, A.d, 0 - lam + dr, xcur);
            for (int t = 0; t < TF; ++t) {
                double cin[LQ + 2], xnext[DP];
                cin[0] = 0.0;
#pragma unroll
                for (int m = 1; m < LQ + 2; ++m) cin[m] = wave_from_left<G>(fw.sout[m]);
                const int a = t - lam;
                wave_load_point<DP>(A.X, i, A.L1, A.d, a + 1 + dr, xnext);
                if (a >= 0 && a < R1) {
                    double dm[C];
                    dmg.row(xcur, true, A.kind, A.p0, A.p1, dm);
                    fw.step(dm, cin, M);
#pragma unroll
                    for (int m = 0; m < LQ; ++m)
                        if (m < M - 1) {
#pragma unroll
                            for (int c = 0; c < C; ++c) slot(m, t, lam, c) = fw.q[m][c];
                        }
                }
#pragma unroll
                for (int f = 0; f < DP; ++f) xcur[f] = xnext[f];
            }
        }
        __threadfence();        // the backward sweep reads what other lanes of this wavefront stored

        // ---- backward sweep
        {
            WaveBwd<C, LQ> bw;
            bw.reset();
            if (MODE != MODE_PT_NODIFF) {
                double xl[DP];
                wave_load_point<DP>(A.X, i, A.L1, A.d, R1, xl);
                dmg.prime(xl, A.kind, A.p0, A.p1);
            }
            double* lamrow = A.lam + size_t(have ? pp : 0) * R1 * R2;
            double xcur[DP], qcur[LQ][C];
            wave_load_point<DP>(A.X, i, A.L1, A.d, R1 - 1 + (G - 1 - lam), xcur);     // row of step 0 (beyond the sequence: zeros)
            // forward values Q_m[a-1][b_c - 1] of lattice row a: written at forward step a - 1 + lam (own columns) and one
            // step earlier by the left neighbour (first column); fetched one backward step ahead of their use
            auto fetch_q = [&](int a, double (&q)[LQ][C]) {
                const int tf = a - 1 + lam;
#pragma unroll
                for (int m = 0; m < LQ; ++m)
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        double v = 0.0;
                        if (m < M - 1 && a > 0 && a < R1) {
                            if (c > 0) v = slot(m, tf, lam, c - 1);
                            else if (lam > 0) v = slot(m, tf - 1, lam - 1, C - 1);
                        }
                        q[m][c] = v;
                    }
            };
            fetch_q(R1 - 1 + (G - 1 - lam), qcur);
            for (int u = 0; u < TF; ++u) {
                double sin[LQ], xnext[DP], qnext[LQ][C];
#pragma unroll
                for (int p = 0; p < LQ; ++p) sin[p] = wave_from_right<G>(bw.svout[p]);
                const int a = R1 - 1 - (u - (G - 1 - lam));
                wave_load_point<DP>(A.X, i, A.L1, A.d, a - 1, xnext);
                fetch_q(a - 1, qnext);
                if (a >= 0 && a < R1) {
                    double dm[C], lv[C];
                    dmg.row(xcur, false, A.kind, A.p0, A.p1, dm);
                    bw.step(dm, clev, qcur, sin, M, lv);
                    if (have) {
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            if (c < dmg.nvalid) lamrow[size_t(a) * R2 + C * lam + c] = lv[c];
                    }
                }
#pragma unroll
                for (int f = 0; f < DP; ++f) xcur[f] = xnext[f];
#pragma unroll
                for (int m = 0; m < LQ; ++m)
#pragma unroll
                    for (int c = 0; c < C; ++c) qcur[m][c] = qnext[m][c];
            }
        }
        __threadfence();        // the slot is rewritten by the next pair
    }
}

// One launch contracts Lam of a block of pairs into the gradient of ONE side.
//   SIDE 0: target = x.  grid (target sequences, partner slices), threads = target points p; inner loops: partner sequences, partner points q.
//   SIDE 1: target = y.  Same with the roles (and the indices of Lam) exchanged.
// Lam of pair (i, j) sits at lam + ((i - i0) * nj + (j - j0)) * R1 * R2 (diag: pair i at (i - i0) * R1 * R2).
struct LamContractArgs {
    const double* X; const double* Y;      // (N, L, d) row-major scaled observations
    int L1, L2, d, kind, mode;
    double p0, p1;
    const double* lam;
    int64_t i0, ni, j0, nj;                // block of pairs covered by lam (diag: j == i, nj ignored)
    int diag;
    double* gT;                            // gradient of the target side, user layout (N, L, d), accumulated with atomics
    double* gbase;                         // optional; only SIDE 0 adds to it
    int nslices;                           // partner sequences are dealt round-robin to gridDim.y slices
};

// KIND >= 0 fixes the base kernel at compile time (-1: A.kind); NODIFF: MODE_PT_NODIFF (Gam = Lam).
template <int DP, int SIDE, int KIND, bool NODIFF>
__global__ void __launch_bounds__(256) lam_contract_kernel(const LamContractArgs A) {
    extern __shared__ double part[];       // one partner sequence: Lp x DP, then its Lp squared norms
    constexpr int dr = NODIFF ? 0 : 1;
    const int kind = KIND >= 0 ? KIND : A.kind;
    const int R1 = A.L1 - dr, R2 = A.L2 - dr;
    const int Lt = SIDE == 0 ? A.L1 : A.L2, Lp = SIDE == 0 ? A.L2 : A.L1;
    const int64_t tseq = (SIDE == 0 ? A.i0 : A.j0) + blockIdx.x;          // target sequence
    const double* T = SIDE == 0 ? A.X : A.Y;
    const double* P = SIDE == 0 ? A.Y : A.X;
    const int64_t np = A.diag ? 1 : (SIDE == 0 ? A.nj : A.ni);
    constexpr bool nodiff = NODIFF;
    double* pnorm = part + Lp * DP;
    double gp0 = 0.0;
    for (int tp0 = 0; tp0 < Lt; tp0 += blockDim.x) {                      // target points in tiles of blockDim.x
        const int tp = tp0 + threadIdx.x;
        const bool tv = tp < Lt;
        double xt[DP], acc[DP], accs = 0.0, ts = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f) {
            xt[f] = (tv && f < A.d) ? T[(tseq * Lt + tp) * A.d + f] : 0.0;
            acc[f] = 0.0;
            ts = fma(xt[f], xt[f], ts);
        }
        for (int64_t k = blockIdx.y; k < np; k += A.nslices) {
            const int64_t pseq = A.diag ? tseq : (SIDE == 0 ? A.j0 : A.i0) + k;
            __syncthreads();
            for (int e = threadIdx.x; e < Lp * DP; e += blockDim.x) {
                const int q = e / DP, f = e % DP;
                part[e] = f < A.d ? P[(pseq * Lp + q) * A.d + f] : 0.0;
            }
            __syncthreads();
            for (int q = threadIdx.x; q < Lp; q += blockDim.x) {
                double ps = 0.0;
#pragma unroll
                for (int f = 0; f < DP; ++f) ps = fma(part[q * DP + f], part[q * DP + f], ps);
                pnorm[q] = ps;
            }
            __syncthreads();
            const int64_t pi = A.diag ? (tseq - A.i0) : (SIDE == 0 ? (tseq - A.i0) * A.nj + k : k * A.nj + (tseq - A.j0));
            const double* lm = A.lam + pi * int64_t(R1) * R2;
            if (tv) {
                // Gam[p][q] = Lam[p-1][q-1] - Lam[p-1][q] - Lam[p][q-1] + Lam[p][q]  (zero outside the lattice); nodiff: Gam = Lam
                // SIDE 0: p = tp fixed, q runs; SIDE 1: q = tp fixed, p runs.
                double lo_prev = 0.0, hi_prev = 0.0;      // Lam at (fixed-1, run-1), (fixed, run-1)
                const int Rr = SIDE == 0 ? R2 : R1, Rf = SIDE == 0 ? R1 : R2;
                const bool wave_first = (threadIdx.x & 63) == 0;
                constexpr int NU = 8;
                auto lam_at = [&](int fixed, int run) -> double { return SIDE == 0 ? lm[int64_t(fixed) * R2 + run] : lm[int64_t(run) * R2 + fixed]; };
                for (int r0 = 0; r0 < Lp; r0 += NU) {
                    // NU cells of the running index at a time: their loads are in flight together, and the neighbouring
                    // value across the fixed index is the previous lane's (one load per cell instead of two)
                    double hi4[NU], lo4[NU];
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int r = r0 + u;
                        if (nodiff) hi4[u] = r < Lp ? lam_at(tp, r) : 0.0;
                        else hi4[u] = (r < Rr && tp < Rf) ? lam_at(tp, r) : 0.0;
                    }
                    if (!nodiff) {
#pragma unroll
                        for (int u = 0; u < NU; ++u) {
                            const int r = r0 + u;
                            double lo = __shfl_up(hi4[u], 1, 64);        // lane tp - 1 holds Lam(tp - 1, r) as its own cell
                            if (wave_first) lo = (tp > 0 && r < Rr) ? lam_at(tp - 1, r) : 0.0;
                            lo4[u] = lo;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int r = r0 + u;
                        if (r >= Lp) break;
                        double gam;
                        if (nodiff) {
                            gam = hi4[u];
                        } else {
                            gam = (lo_prev - lo4[u]) - (hi_prev - hi4[u]);
                            lo_prev = lo4[u];
                            hi_prev = hi4[u];
                        }
                        const double* yq = part + r * DP;
                        double in = 0.0;
                        const double ps = pnorm[r];
#pragma unroll
                        for (int f = 0; f < DP; ++f) in = fma(xt[f], yq[f], in);
                        // derivative with respect to the target point; base_eval_grad's first argument is x
                        const BaseGrad g = SIDE == 0 ? base_eval_grad(kind, in, ts, ps, A.p0, A.p1) : base_eval_grad(kind, in, ps, ts, A.p0, A.p1);
                        const double w = gam * (g.cy - g.cd);
                        accs = fma(gam, (SIDE == 0 ? g.cx : g.cx2) + g.cd, accs);
                        if (SIDE == 0) gp0 = fma(gam, g.dp0, gp0);
#pragma unroll
                        for (int f = 0; f < DP; ++f) acc[f] = fma(w, yq[f], acc[f]);
                    }
                }
            }
        }
        if (tv) {
#pragma unroll
            for (int f = 0; f < DP; ++f)
                if (f < A.d) atomicAdd(&A.gT[(tseq * Lt + tp) * A.d + f], fma(accs, xt[f], acc[f]));
        }
    }
    if (SIDE == 0 && A.gbase) {
        grad_add(&A.gbase[0], gp0, true, true);          // one atomic per wavefront
    }
}

}  // namespace gpsig

namespace gpsig {

// ---- scratch-free kernel (WaveUndo / WaveGy) ------------------------------------------------------------------------------
// A task = one register-side sequence against a run of streamed sequences; 64/G tasks per wavefront in lock step.
struct Wave2Args {
    const double* S; const double* R;      // streamed side / register-resident side, user layout (N, L, d)
    double* gR;                            // gradient of the register-resident side, (NR, LR, d), accumulated with atomics
    int NS, NR, LS, LR, d;
    int M, kind, mode;
    double p0, p1;
    const SeqTask* tasks;                  // y0 = register-side sequence, x0 / nx = run of streamed sequences
    int ntasks;
    const double* G; int64_t gm, gs, gr;   // upstream: G[m * gm + s * gs + r * gr]
    int gsym;                              // add the transposed term G[m * gm + r * gs + s * gr] (symmetric Gram)
    double gscale;                         // 1, or 2 for the diagonal (both roles of the same sequence)
    double* gbase; double gbase_scale;     // optional: d/d base_params[0], scaled (0.5 where both orders of a pair are swept)
    // seq_lam_undo_kernel only: tasks index a block of pairs, streamed i0 .. i0+ni against register-side j0 .. j0+nj
    double* lam;                           // Lam of pair (i, j) at ((i - i0) * nj + (j - j0)) * R1 * R2 (diag: (i - i0) * R1 * R2)
    int64_t i0, j0, nj;
    int diag;
    int64_t gsym_from;                     // gsym: the transposed term is added for register-side sequences >= gsym_from only
};

// dynamic LDS: (64 / G) * (LS rows) * LQ doubles.  MX: num_levels == LQ + 1 at compile time (no level predicates in the sweeps).
template <int G, int C, int DP, int LQ, int MODE, bool MX = false>
__global__ void __launch_bounds__(64) seq_grad_wave2_kernel(const Wave2Args A) {
    extern __shared__ double w2_sm[];
    constexpr int PW = 64 / G;
    const int lane = threadIdx.x, lam = lane % G, gw = lane / G;
    const int dr = MODE == MODE_PT_NODIFF ? 0 : 1;
    const int R1 = A.LS - dr, R2 = A.LR - dr, M = MX ? LQ + 1 : A.M;
    const int TF = R1 + G - 1;
    double* rt = w2_sm + size_t(gw) * (R1 > 0 ? R1 : 1) * LQ;         // rowtot[a][m-1]
    const int tid = blockIdx.x * PW + gw;
    const bool has_task = tid < A.ntasks;
    const SeqTask tk = has_task ? A.tasks[tid] : SeqTask{0, 0, 0};
    // the longest run in this wavefront
    int nmax = tk.nx;
#pragma unroll
    for (int o = G; o < 64; o <<= 1) { const int v = __shfl_xor(nmax, o, 64); nmax = v > nmax ? v : nmax; }
    const int64_t r = tk.y0;
    const int last_lane = R2 > 0 ? (R2 - 1) / C : 0;

    WaveGy<C, DP, MODE> gy;
    {
        double ypts[C + 1][DP];
#pragma unroll
        for (int c = 0; c <= C; ++c) wave_load_point<DP>(A.R, r, A.LR, A.d, C * lam + c, ypts[c]);
        int nv = R2 - C * lam;
        gy.set_y(ypts, nv < 0 ? 0 : (nv > C ? C : nv));
    }
    for (int it = 0; it < nmax; ++it) {
        const bool have = has_task && it < tk.nx;
        const int64_t s = tk.x0 + (have ? it : 0);
        double clev[LQ + 2];
#pragma unroll
        for (int p = 0; p < LQ + 2; ++p) {
            double v = 0.0;
            if (have && p >= 1 && p <= M) {
                v = A.G[p * A.gm + s * A.gs + r * A.gr];
                if (A.gsym) v += A.G[p * A.gm + r * A.gs + s * A.gr];
                v *= A.gscale;
            }
            clev[p] = v;
        }
        WaveFwd<C, LQ> fw;
        fw.reset();
        double xn[DP];
        if (MODE != MODE_PT_NODIFF) {
            wave_load_point<DP>(A.S, s, A.LS, A.d, 0, xn);
            gy.prime_fwd(xn, A.kind, A.p0, A.p1);
        }
        wave_load_point<DP>(A.S, s, A.LS, A.d, 0 - lam + dr, xn);          // row of step 0
        for (int t = 0; t < TF; ++t) {
            double cin[LQ + 2], xnext[DP];
            cin[0] = 0.0;
#pragma unroll
            for (int m = 1; m < LQ + 2; ++m) cin[m] = wave_from_left<G>(fw.sout[m]);
            const int a = t - lam;
            wave_load_point<DP>(A.S, s, A.LS, A.d, a + 1 + dr, xnext);     // prefetch the row of step t+1
            if (a >= 0 && a < R1) {
                double dm[C];
                gy.row_fwd(xn, A.kind, A.p0, A.p1, dm);
                fw.step(dm, cin, M);
                if (lam == last_lane) {
#pragma unroll
                    for (int m = 1; m <= LQ; ++m) rt[a * LQ + m - 1] = m < M ? fw.sout[m] : 0.0;
                }
            }
#pragma unroll
            for (int f = 0; f < DP; ++f) xn[f] = xnext[f];
        }
        __syncthreads();
        WaveUndo<C, LQ> bw;
        bw.init(fw);
        {
            double xl[DP];
            wave_load_point<DP>(A.S, s, A.LS, A.d, R1, xl);
            gy.prime(xl, A.kind, A.p0, A.p1);
        }
        wave_load_point<DP>(A.S, s, A.LS, A.d, R1 - 1 + (G - 1 - lam), xn);  // row of step 0 (out of range -> zeros)
        for (int u = 0; u < TF; ++u) {
            double sufin[LQ], svin[LQ], xnext[DP];
#pragma unroll
            for (int p = 0; p < LQ; ++p) {
                sufin[p] = wave_from_right<G>(bw.sufout[p]);
                svin[p] = wave_from_right<G>(bw.svout[p]);
            }
            const int a = R1 - 1 - (u - (G - 1 - lam));
            wave_load_point<DP>(A.S, s, A.LS, A.d, a - 1, xnext);
            if (a >= 0 && a < R1) {
                double dm[C], rtv[LQ], lv[C];
#pragma unroll
                for (int p = 0; p < LQ; ++p) rtv[p] = rt[a * LQ + p];
                gy.row(xn, A.kind, A.p0, A.p1, dm);
                bw.step(dm, clev, rtv, sufin, svin, M, a == 0, lam == 0, lv);
                gy.contract(lv);
                if (a == 0) gy.finish_pair();
            }
#pragma unroll
            for (int f = 0; f < DP; ++f) xn[f] = xnext[f];
        }
        __syncthreads();
    }
    if (has_task) {
        const int npts = MODE == MODE_PT_NODIFF ? gy.nvalid : (gy.nvalid > 0 ? gy.nvalid + 1 : 0);
#pragma unroll
        for (int c = 0; c <= C; ++c)
            if (c < npts) {
                const int q = C * lam + c;
#pragma unroll
                for (int f = 0; f < DP; ++f)
                    if (f < A.d) atomicAdd(&A.gR[(r * A.LR + q) * A.d + f], gy.point_grad(c, f));
            }
    }
    if (A.gbase) grad_add(&A.gbase[0], gy.gp0 * A.gbase_scale, true, has_task);
}

// ---- scratch-free forward state, Lam out (WaveUndo / WaveDm): the point kernels ------------------------------------------------
// The sweeps of seq_grad_wave2_kernel with the kernel derivative left out: the backward sweep stores Lam of every pair, and
// lam_contract_kernel turns it into the gradient of both sides.  Per lane this keeps the points of C + 1 columns and the
// recursion state only, so the 16-lane shapes that hold 4 pairs per wavefront fit for every base kernel.
// KIND >= 0 fixes the base kernel at compile time: the C + 1 evaluations of a row then interleave instead of queueing behind a
// switch (1024 x 1024 RBF Gram: 56 -> 34 ms for the sweeps).
#ifndef LAM_UNDO_WAVES
#define LAM_UNDO_WAVES 1        // wavefronts per SIMD the sweeps are compiled for.  2: the 256 registers cost 12 scratch accesses per step of either
                                // sweep -- K(X) forward + backward of 1,024 sequences at the headline shape 48.3 -> 77.4 ms (round 5, same box).
                                // Also tried there: the sweeps' five kernel values per row through the table-driven exp in hand-scheduled pairs
                                // (exp_pair_asm.hpp): 20 % fewer vector instructions (274 -> 214 / 375 -> 328 per step) and SLOWER, 48.8 -> 51.1 ms --
                                // at one wavefront per SIMD the table reads' latency is exposed where the library exps interleave (68.7 ms with two
                                // waves).  The sweeps need a per-lane state that fits two wavefronts per SIMD before anything else pays.
#endif
template <int G, int C, int DP, int LQ, int MODE, bool MX = false, int KIND = -1>
__global__ void __launch_bounds__(64, LAM_UNDO_WAVES) seq_lam_undo_kernel(const Wave2Args A) {
    extern __shared__ double w2_sm[];
    const int kind = KIND >= 0 ? KIND : A.kind;
    constexpr int PW = 64 / G;
    const int lane = threadIdx.x, ln = lane % G, gw = lane / G;
    const int dr = MODE == MODE_PT_NODIFF ? 0 : 1;
    const int R1 = A.LS - dr, R2 = A.LR - dr, M = MX ? LQ + 1 : A.M;
    const int TF = R1 + G - 1;
    double* rt = w2_sm + size_t(gw) * (R1 > 0 ? R1 : 1) * LQ;         // rowtot[a][m-1]
    const int tid = blockIdx.x * PW + gw;
    const bool has_task = tid < A.ntasks;
    const SeqTask tk = has_task ? A.tasks[tid] : SeqTask{0, 0, 0};
    int nmax = tk.nx;
#pragma unroll
    for (int o = G; o < 64; o <<= 1) { const int v = __shfl_xor(nmax, o, 64); nmax = v > nmax ? v : nmax; }
    const int64_t r = A.j0 + tk.y0;
    const int last_lane = R2 > 0 ? (R2 - 1) / C : 0;

    WaveDm<C, DP, MODE> dmg;
    {
        double ypts[C + 1][DP];
#pragma unroll
        for (int c = 0; c <= C; ++c) wave_load_point<DP>(A.R, r, A.LR, A.d, C * ln + c, ypts[c]);
        int nv = R2 - C * ln;
        dmg.set_y(ypts, nv < 0 ? 0 : (nv > C ? C : nv));
    }
    for (int it = 0; it < nmax; ++it) {
        const bool have = has_task && it < tk.nx;
        const int64_t sl = tk.x0 + (have ? it : 0), s = A.i0 + sl;
        double clev[LQ + 2];
#pragma unroll
        for (int p = 0; p < LQ + 2; ++p) {
            double v = 0.0;
            if (have && p >= 1 && p <= M) {
                v = A.G[p * A.gm + s * A.gs + r * A.gr];
                if (A.gsym && r >= A.gsym_from) v += A.G[p * A.gm + r * A.gs + s * A.gr];
            }
            clev[p] = v;
        }
        WaveFwd<C, LQ> fw;
        fw.reset();
        double xn[DP];
        if (MODE != MODE_PT_NODIFF) {
            wave_load_point<DP>(A.S, s, A.LS, A.d, 0, xn);
            dmg.prime(xn, kind, A.p0, A.p1);
        }
        wave_load_point<DP>(A.S, s, A.LS, A.d, 0 - ln + dr, xn);           // row of step 0
        for (int t = 0; t < TF; ++t) {
            double cin[LQ + 2], xnext[DP];
            cin[0] = 0.0;
#pragma unroll
            for (int m = 1; m < LQ + 2; ++m) cin[m] = wave_from_left<G>(fw.sout[m]);
            const int a = t - ln;
            wave_load_point<DP>(A.S, s, A.LS, A.d, a + 1 + dr, xnext);     // prefetch the row of step t+1
            if (a >= 0 && a < R1) {
                double dm[C];
                dmg.row(xn, true, kind, A.p0, A.p1, dm);
                fw.step(dm, cin, M);
                if (ln == last_lane) {
#pragma unroll
                    for (int m = 1; m <= LQ; ++m) rt[a * LQ + m - 1] = m < M ? fw.sout[m] : 0.0;
                }
            }
#pragma unroll
            for (int f = 0; f < DP; ++f) xn[f] = xnext[f];
        }
        __syncthreads();
        WaveUndo<C, LQ> bw;
        bw.init(fw);
        if (MODE != MODE_PT_NODIFF) {
            double xl[DP];
            wave_load_point<DP>(A.S, s, A.LS, A.d, R1, xl);
            dmg.prime(xl, kind, A.p0, A.p1);
        }
        double* lamrow = A.lam + size_t(A.diag ? sl : sl * A.nj + tk.y0) * R1 * R2;
        wave_load_point<DP>(A.S, s, A.LS, A.d, R1 - 1 + (G - 1 - ln), xn);   // row of step 0 (out of range -> zeros)
        for (int u = 0; u < TF; ++u) {
            double sufin[LQ], svin[LQ], xnext[DP];
#pragma unroll
            for (int p = 0; p < LQ; ++p) {
                sufin[p] = wave_from_right<G>(bw.sufout[p]);
                svin[p] = wave_from_right<G>(bw.svout[p]);
            }
            const int a = R1 - 1 - (u - (G - 1 - ln));
            wave_load_point<DP>(A.S, s, A.LS, A.d, a - 1, xnext);
            if (a >= 0 && a < R1) {
                double dm[C], rtv[LQ], lv[C];
#pragma unroll
                for (int p = 0; p < LQ; ++p) rtv[p] = rt[a * LQ + p];
                dmg.row(xn, false, kind, A.p0, A.p1, dm);
                bw.step(dm, clev, rtv, sufin, svin, M, a == 0, ln == 0, lv);
                if (have) {
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        if (c < dmg.nvalid) lamrow[size_t(a) * R2 + C * ln + c] = lv[c];
                }
            }
#pragma unroll
            for (int f = 0; f < DP; ++f) xn[f] = xnext[f];
        }
        __syncthreads();
    }
}

}  // namespace gpsig
