// grad_wave_ho_kernel.hpp -- reverse pass of the HIGHER-ORDER sequence-vs-sequence recursion (signature_kern_higher_order,
// gpsig/signature_algs.py:37-74) as two skewed sweeps of a wavefront over the lattice of one pair: the fused form of the lattice
// operations of grad_ho_kernels.hpp (one launch per pair block instead of ~250 passes over HBM-resident lattices).
//
// The recursion, cell by cell.  With M[a][b] the (double-increment) lattice of base-kernel values and d_i = min(i, order), level i holds
// a d_i x d_i grid of lattices R_i[r][k] (r, k = repeat counts along the two time axes, signature_algs.py:60-69):
//     R_1[0][0]     = M
//     R_{i+1}[0][0] = M * P_i          P_i [a][b]    = sum_{a' < a, b' < b} tot_i[a'][b'],    tot_i = sum_{r,k} R_i[r][k]          (:64)
//     R_{i+1}[0][k] = M/(k+1) * CP_i[k-1]   CP_i[k][a][b] = sum_{a' < a} col_i[k][a'][b],     col_i[k] = sum_r R_i[r][k]           (:66)
//     R_{i+1}[r][0] = M/(r+1) * RP_i[r-1]   RP_i[r][a][b] = sum_{b' < b} row_i[r][a][b'],     row_i[r] = sum_k R_i[r][k]           (:67)
//     R_{i+1}[r][k] = M/((r+1)(k+1)) * R_i[r-1][k-1]                                                                             (:69)
//     K_i = sum_cells tot_i                                                                                                      (:71)
// Reverse mode, with c_i the upstream gradient of K_i and g_i[r][k] = dL/dR_i[r][k] at a cell (g_M == c_M):
//     g_i[r][k] = c_i + SP_i + SCP_i[k] + SRP_i[r] + M/((r+2)(k+2)) g_{i+1}[r+1][k+1]          (terms that exist in level i+1's grid only)
//     SP_i [a][b]    = sum_{a' > a, b' > b} M g_{i+1}[0][0]           SCP_i[k][a][b] = sum_{a' > a} M/(k+2) g_{i+1}[0][k+1]  (same column)
//     SRP_i[r][a][b] = sum_{b' > b} M/(r+2) g_{i+1}[r+1][0]  (same row)
//     Lam[a][b] = dL/dM[a][b] = sum_i ( g_i[0][0] P_{i-1} + sum_k g_i[0][k] CP_{i-1}[k-1]/(k+1) + sum_r g_i[r][0] RP_{i-1}[r-1]/(r+1)
//                                      + sum_{r,k >= 1} g_i[r][k] R_{i-1}[r-1][k-1]/((r+1)(k+1)) ),      P_0 == 1.
// Mapping: as grad_wave_kernel.hpp -- a pair occupies G consecutive lanes, lane lam owns C lattice columns and works on row t - lam at
// step t of the forward sweep (row prefixes handed to the right neighbour by one DPP shift per word and step), rows descending with the
// opposite skew in the backward sweep (row suffixes to the left neighbour).  The forward sweep leaves, per cell, the prefixes the cell
// READ (P_j, CP_j[.], RP_j[.], j < M: 3 (M-1) words at order 2) in an HBM slot of the pair group, indexed by the step so that a
// wavefront's stores are contiguous; the backward sweep reads them back in the same lane (no cross-lane dependence through memory),
// rebuilds the cell's grids R_1 .. R_{M-1} from them, and runs the adjoints down the levels.  M[a][b] itself comes from an HBM lattice
// (ho_dm_kernel): the sweeps are independent of the base kernel and of the number of feature columns; Lam goes to lam_contract_kernel.
// ORDER is a compile-time parameter (2, 3, 4: grid shapes known at every level), num_levels a run-time one (<= LQ + 1).
#pragma once

#include <type_traits>
#include <utility>

#include "grad_wave_kernel.hpp"

namespace gpsig {

struct WaveHoArgs {
    const double* dM;                      // (npairs, R1, R2) row-major
    const double* G; int64_t gm, gi, gj;   // upstream gradient of the levels: G[p * gm + i * gi + j * gj]
    int N2, diag;
    int64_t pair0, npairs;                 // pairs pair0 .. pair0 + npairs - 1 of the enumeration p = i * N2 + j (diag: p = i)
    int R1, R2, M;
    double* scratch;                       // per group slot: ho_stash_words(M) * (R1 + G - 1) * G * C doubles
    double* lam;                           // out: (npairs, R1, R2)
    int ngroups;
};

template <int O> constexpr int ho_dim(int i) { return i < O ? i : O; }              // d_i
template <int O> constexpr int ho_nk(int j) { return ho_dim<O>(j + 1) - 1; }         // column / row prefixes of level j that level j + 1 reads
template <int O> constexpr int ho_stash_off(int j) {                                 // words of the levels below j: P, CP[.], RP[.] each
    int s = 0;
    for (int i = 1; i < j; ++i) s += 1 + 2 * ho_nk<O>(i);
    return s;
}
inline int ho_stash_words(int order, int M) {                                       // host side: words per cell for num_levels M
    int s = 0;
    for (int i = 1; i < M; ++i) s += 1 + 2 * (((i + 1) < order ? (i + 1) : order) - 1);
    return s;
}

template <int B, int E, class F>
__device__ __forceinline__ void ho_static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        ho_static_for<B + 1, E>(f);
    }
}
template <int B, int E, class F>
__device__ __forceinline__ void ho_static_for_down(F&& f) {          // B, B-1, .., E+1
    if constexpr (B > E) {
        f(std::integral_constant<int, B>{});
        ho_static_for_down<B - 1, E>(f);
    }
}

// the grid of level J+1 at one cell from level J's grid and the prefixes the cell reads
template <int O, int J>
__device__ __forceinline__ void ho_next_grid(double dm, double P, const double (&CP)[O - 1], const double (&RP)[O - 1], const double (&Rp)[O][O],
                                             double (&Rn)[O][O]) {
    constexpr int dn = ho_dim<O>(J + 1);
    Rn[0][0] = dm * P;
#pragma unroll
    for (int k = 1; k < dn; ++k) {
        Rn[0][k] = (dm * (1.0 / double(k + 1))) * CP[k - 1];
        Rn[k][0] = (dm * (1.0 / double(k + 1))) * RP[k - 1];
    }
#pragma unroll
    for (int r = 1; r < dn; ++r)
#pragma unroll
        for (int k = 1; k < dn; ++k) Rn[r][k] = (dm * (1.0 / double((r + 1) * (k + 1)))) * Rp[r - 1][k - 1];
}

// ---- forward sweep state of one lane -------------------------------------------------------------------------------------------------
template <int C, int LQ, int O>
struct WaveHoFwd {
    double q[LQ][C], qg[LQ];        // sum of tot_j over rows <= the previous one and columns <= b_c; the ghost column b_0 - 1
    double cp[LQ][O - 1][C];        // sum of col_j[k] over the earlier rows of the lane's columns
    double st[LQ], sr[LQ][O - 1];   // end-of-chunk row prefixes of tot_j, row_j[k]: what the right neighbour takes over

    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int j = 0; j < LQ; ++j) {
            qg[j] = st[j] = 0.0;
#pragma unroll
            for (int c = 0; c < C; ++c) q[j][c] = 0.0;
#pragma unroll
            for (int k = 0; k < O - 1; ++k) {
                sr[j][k] = 0.0;
#pragma unroll
                for (int c = 0; c < C; ++c) cp[j][k][c] = 0.0;
            }
        }
    }
    // ct, cr: the left neighbour's st, sr of ITS previous step (zeros for the first lane of a pair).  store(word, column, value).
    template <class Store>
    __device__ __forceinline__ void step(const double (&dm)[C], const double (&ct)[LQ], const double (&cr)[LQ][O - 1], int M, Store&& store) {
        double none = 0.0;
        step_impl<false>(dm, ct, cr, M, store, none);
    }
    // TOP (M == LQ + 1): the cells' totals of level M's grid are added to ktop (the forward pass proper: K_M = their sum over the lattice)
    template <bool TOP, class Store>
    __device__ __forceinline__ void step_impl(const double (&dm)[C], const double (&ct)[LQ], const double (&cr)[LQ][O - 1], int M, Store&& store, double& ktop) {
        double pv[LQ][C], rt[LQ], rr[LQ][O - 1];
#pragma unroll
        for (int j = 0; j < LQ; ++j) {
            rt[j] = ct[j];
#pragma unroll
            for (int k = 0; k < O - 1; ++k) rr[j][k] = cr[j][k];
#pragma unroll
            for (int c = 0; c < C; ++c) pv[j][c] = c == 0 ? qg[j] : q[j][c > 0 ? c - 1 : 0];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            double Rp[O][O];
            Rp[0][0] = dm[c];
            ho_static_for<1, LQ + 1>([&](auto jc) {
                constexpr int J = decltype(jc)::value, dj = ho_dim<O>(J), nk = ho_nk<O>(J), off = ho_stash_off<O>(J);
                if (J < M) {
                    double tot = 0.0, col[O - 1], row[O - 1];
#pragma unroll
                    for (int r = 0; r < dj; ++r)
#pragma unroll
                        for (int k = 0; k < dj; ++k) tot += Rp[r][k];
#pragma unroll
                    for (int k = 0; k < nk; ++k) {
                        double sc = 0.0, sw = 0.0;
#pragma unroll
                        for (int r = 0; r < dj; ++r) { sc += Rp[r][k]; sw += Rp[k][r]; }
                        col[k] = sc; row[k] = sw;
                    }
                    double CP[O - 1], RP[O - 1];
#pragma unroll
                    for (int k = 0; k < O - 1; ++k) { CP[k] = k < nk ? cp[J - 1][k][c] : 0.0; RP[k] = k < nk ? rr[J - 1][k] : 0.0; }
                    store(off, c, pv[J - 1][c]);
#pragma unroll
                    for (int k = 0; k < nk; ++k) { store(off + 1 + k, c, CP[k]); store(off + 1 + nk + k, c, RP[k]); }
                    if constexpr (J < LQ) {
                        double Rn[O][O];
                        ho_next_grid<O, J>(dm[c], pv[J - 1][c], CP, RP, Rp, Rn);
                        constexpr int dn = ho_dim<O>(J + 1);
#pragma unroll
                        for (int r = 0; r < dn; ++r)
#pragma unroll
                            for (int k = 0; k < dn; ++k) Rp[r][k] = Rn[r][k];
                    } else if constexpr (TOP) {
                        double Rn[O][O];
                        ho_next_grid<O, J>(dm[c], pv[J - 1][c], CP, RP, Rp, Rn);
                        constexpr int dn = ho_dim<O>(J + 1);
#pragma unroll
                        for (int r = 0; r < dn; ++r)
#pragma unroll
                            for (int k = 0; k < dn; ++k) ktop += Rn[r][k];
                    }
                    rt[J - 1] += tot;
                    q[J - 1][c] += rt[J - 1];
#pragma unroll
                    for (int k = 0; k < nk; ++k) { cp[J - 1][k][c] += col[k]; rr[J - 1][k] += row[k]; }
                }
            });
        }
#pragma unroll
        for (int j = 0; j < LQ; ++j) {
            st[j] = rt[j];
            if (j + 1 < M) qg[j] += ct[j];
#pragma unroll
            for (int k = 0; k < O - 1; ++k) sr[j][k] = rr[j][k];
        }
    }
};

// ---- backward sweep state of one lane ------------------------------------------------------------------------------------------------
template <int C, int LQ, int O>
struct WaveHoBwd {
    double qb[LQ][C], qbg[LQ];      // level p: sum of M g_{p+1}[0][0] over rows > a and columns >= b_c; the ghost column b_{C-1} + 1
    double scp[LQ][O - 1][C];       // sum of M/(k+2) g_{p+1}[0][k+1] over the later rows of the lane's columns
    double sv[LQ], sw[LQ][O - 1];   // row suffixes handed to the left neighbour

    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int p = 0; p < LQ; ++p) {
            qbg[p] = sv[p] = 0.0;
#pragma unroll
            for (int c = 0; c < C; ++c) qb[p][c] = 0.0;
#pragma unroll
            for (int k = 0; k < O - 1; ++k) {
                sw[p][k] = 0.0;
#pragma unroll
                for (int c = 0; c < C; ++c) scp[p][k][c] = 0.0;
            }
        }
    }
    // clev[i], i = 1..M: upstream gradients.  iv, iw: the right neighbour's sv, sw of ITS previous step.  fetch(word, column): what the
    // forward sweep stored for this row.
    template <class Fetch>
    __device__ __forceinline__ void step(const double (&dm)[C], const double (&clev)[LQ + 2], const double (&iv)[LQ], const double (&iw)[LQ][O - 1],
                                         int M, Fetch&& fetch, double (&lam)[C]) {
        double spv[LQ][C], rv[LQ], rw[LQ][O - 1];
#pragma unroll
        for (int p = 0; p < LQ; ++p) {
            rv[p] = iv[p];
#pragma unroll
            for (int k = 0; k < O - 1; ++k) rw[p][k] = iw[p][k];
#pragma unroll
            for (int c = 0; c < C; ++c) spv[p][c] = c < C - 1 ? qb[p][c < C - 1 ? c + 1 : c] : qbg[p];
        }
#pragma unroll
        for (int c = C - 1; c >= 0; --c) {
            // the cell's forward side: prefixes from the slot, the grids of levels 1 .. M-1 rebuilt from them
            double P[LQ], CP[LQ][O - 1], RP[LQ][O - 1], R[LQ][O][O];
            R[0][0][0] = dm[c];
            ho_static_for<1, LQ + 1>([&](auto jc) {
                constexpr int J = decltype(jc)::value, nk = ho_nk<O>(J), off = ho_stash_off<O>(J);
                const bool on = J < M;
                P[J - 1] = on ? fetch(off, c) : 0.0;
#pragma unroll
                for (int k = 0; k < O - 1; ++k) {
                    CP[J - 1][k] = (k < nk && on) ? fetch(off + 1 + (k < nk ? k : 0), c) : 0.0;
                    RP[J - 1][k] = (k < nk && on) ? fetch(off + 1 + nk + (k < nk ? k : 0), c) : 0.0;
                }
                if constexpr (J < LQ) ho_next_grid<O, J>(dm[c], P[J - 1], CP[J - 1], RP[J - 1], R[J - 1], R[J]);
            });
            double l = 0.0, gn[O][O];
#pragma unroll
            for (int r = 0; r < O; ++r)
#pragma unroll
                for (int k = 0; k < O; ++k) gn[r][k] = 0.0;
            ho_static_for_down<LQ + 1, 0>([&](auto ic) {
                constexpr int I = decltype(ic)::value, di = ho_dim<O>(I), dn = ho_dim<O>(I + 1);
                if (I <= M) {
                    double gc[O][O];
                    const bool inner = I < M;                       // level I + 1 exists: its adjoints reach this level
#pragma unroll
                    for (int r = 0; r < di; ++r)
#pragma unroll
                        for (int k = 0; k < di; ++k) {
                            double g = clev[I];
                            if constexpr (I <= LQ) {
                                double e = spv[I - 1][c];
                                if (k + 1 < dn) e += scp[I - 1][k < O - 1 ? k : 0][c];
                                if (r + 1 < dn) e += rw[I - 1][r < O - 1 ? r : 0];
                                if (r + 1 < dn && k + 1 < dn)
                                    e = fma(dm[c] * (1.0 / double((r + 2) * (k + 2))), gn[r + 1 < O ? r + 1 : 0][k + 1 < O ? k + 1 : 0], e);
                                g += inner ? e : 0.0;
                            }
                            gc[r][k] = g;
                        }
                    // this level's share of Lam
                    if constexpr (I == 1) {
                        l += gc[0][0];
                    } else {
                        double s = gc[0][0] * P[I - 2];
#pragma unroll
                        for (int k = 1; k < di; ++k) {
                            s = fma(gc[0][k] * (1.0 / double(k + 1)), CP[I - 2][k - 1], s);
                            s = fma(gc[k][0] * (1.0 / double(k + 1)), RP[I - 2][k - 1], s);
                        }
#pragma unroll
                        for (int r = 1; r < di; ++r)
#pragma unroll
                            for (int k = 1; k < di; ++k) s = fma(gc[r][k] * (1.0 / double((r + 1) * (k + 1))), R[I - 2][r - 1][k - 1], s);
                        l += s;
                    }
                    // level I + 1's adjoints enter the suffix sums of level I (after this cell read them)
                    if constexpr (I <= LQ) {
                        if (inner) {
                            rv[I - 1] = fma(dm[c], gn[0][0], rv[I - 1]);
                            qb[I - 1][c] += rv[I - 1];
#pragma unroll
                            for (int k = 0; k + 1 < dn; ++k) {
                                scp[I - 1][k][c] = fma(dm[c] * (1.0 / double(k + 2)), gn[0][k + 1], scp[I - 1][k][c]);
                                rw[I - 1][k] = fma(dm[c] * (1.0 / double(k + 2)), gn[k + 1][0], rw[I - 1][k]);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < di; ++r)
#pragma unroll
                        for (int k = 0; k < di; ++k) gn[r][k] = gc[r][k];
                }
            });
            lam[c] = l;
        }
#pragma unroll
        for (int p = 0; p < LQ; ++p) {
            sv[p] = rv[p];
            if (p + 1 < M) qbg[p] += iv[p];
#pragma unroll
            for (int k = 0; k < O - 1; ++k) sw[p][k] = rw[p][k];
        }
    }
};

// grid: ngroups / (64 / G) workgroups of one wavefront
template <int G, int C, int LQ, int O>
__global__ void __launch_bounds__(64) seq_grad_wave_ho_kernel(const WaveHoArgs A) {
    constexpr int PW = 64 / G;
    const int lane = threadIdx.x, lam = lane % G;
    const int grp = blockIdx.x * PW + lane / G;
    const int R1 = A.R1, R2 = A.R2, M = A.M;
    const int TF = R1 + G - 1;
    int words = 0;
    ho_static_for<1, LQ + 1>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if (J < M) words = ho_stash_off<O>(J + 1);
    });
    double* const scr = A.scratch + size_t(grp) * size_t(words) * TF * G * C;
    const int64_t rounds = (A.npairs + A.ngroups - 1) / A.ngroups;
    int nvalid = R2 - C * lam;
    nvalid = nvalid < 0 ? 0 : (nvalid > C ? C : nvalid);

    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t pp = rd * A.ngroups + grp;
        const bool have = pp < A.npairs;
        const int64_t pg = A.pair0 + (have ? pp : 0);
        const int64_t i = A.diag ? pg : pg / A.N2, j = A.diag ? pg : pg % A.N2;
        const double* const dmp = A.dM + size_t(have ? pp : 0) * R1 * R2 + C * lam;
        auto load_dm = [&](int a, double (&dm)[C]) {
            const bool ok = a >= 0 && a < R1;
#pragma unroll
            for (int c = 0; c < C; ++c) dm[c] = (ok && c < nvalid) ? dmp[size_t(a) * R2 + c] : 0.0;
        };
        double clev[LQ + 2];
#pragma unroll
        for (int p = 0; p < LQ + 2; ++p) clev[p] = (have && p >= 1 && p <= M) ? A.G[p * A.gm + i * A.gi + j * A.gj] : 0.0;

        // ---- forward sweep
        {
            WaveHoFwd<C, LQ, O> fw;
            fw.reset();
            double dcur[C];
            load_dm(0 - lam, dcur);
            for (int t = 0; t < TF; ++t) {
                double ct[LQ], cr[LQ][O - 1], dnext[C];
#pragma unroll
                for (int m = 0; m < LQ; ++m) {
                    ct[m] = wave_from_left<G>(fw.st[m]);
#pragma unroll
                    for (int k = 0; k < O - 1; ++k) cr[m][k] = wave_from_left<G>(fw.sr[m][k]);
                }
                const int a = t - lam;
                load_dm(a + 1, dnext);
                if (a >= 0 && a < R1)
                    fw.step(dcur, ct, cr, M, [&](int w, int c, double v) { scr[((size_t(w) * TF + t) * G + lam) * C + c] = v; });
#pragma unroll
                for (int c = 0; c < C; ++c) dcur[c] = dnext[c];
            }
        }
        // ---- backward sweep (every lane reads back its own stores: no fence between the sweeps)
        {
            WaveHoBwd<C, LQ, O> bw;
            bw.reset();
            double* const lamrow = A.lam + size_t(have ? pp : 0) * R1 * R2 + C * lam;
            double dcur[C];
            load_dm(R1 - 1 + (G - 1 - lam), dcur);
            for (int u = 0; u < TF; ++u) {
                double iv[LQ], iw[LQ][O - 1], dnext[C];
#pragma unroll
                for (int p = 0; p < LQ; ++p) {
                    iv[p] = wave_from_right<G>(bw.sv[p]);
#pragma unroll
                    for (int k = 0; k < O - 1; ++k) iw[p][k] = wave_from_right<G>(bw.sw[p][k]);
                }
                const int a = R1 - 1 - (u - (G - 1 - lam));
                load_dm(a - 1, dnext);
                if (a >= 0 && a < R1) {
                    double lv[C];
                    const int tf = a + lam;
                    bw.step(dcur, clev, iv, iw, M, [&](int w, int c) { return scr[((size_t(w) * TF + tf) * G + lam) * C + c]; }, lv);
                    if (have) {
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            if (c < nvalid) lamrow[size_t(a) * R2 + c] = lv[c];
                    }
                }
#pragma unroll
                for (int c = 0; c < C; ++c) dcur[c] = dnext[c];
            }
        }
    }
}

// =====================================================================================================================================
// Scratch-free form (the default where its row totals fit LDS).  The backward sweep does not read the prefixes back from memory, it UNDOES
// the forward sweep row by row, level by level within a cell (as WaveUndo does for the first-order recursion, grad_wave_core.hpp):
//     CP_j[k][a][b] = (column sum through row a) - col_j[k][a][b]                                      -- the lane's own accumulator
//     RP_j[k][a][b] = rowtot(row_j[k])[a] - row_j[k][a][b] - (suffix beyond b)                         -- suffixes arrive from the right neighbour
//     P_j[a][b]     = Q_j[a][b-1] - (rowtot(tot_j)[a] - suffix from b on),   Q_j[a][.] undone to Q_j[a-1][.] the same way
// where level j's cell values (tot, col, row) come from its grid, rebuilt from level j-1's prefixes AT THE SAME CELL -- so the undo runs up
// the levels inside a cell, then the adjoints run down.  The forward sweep leaves only the row totals (sum_j (1 + nk_j) words per lattice
// row, in LDS) and every lane's final accumulators (registers).  num_levels and the order are compile-time parameters here: no branches
// around loads, no dead levels in the register file.  HBM traffic per pair: M[a][b] twice, Lam once.
template <int O, int MM> constexpr int ho_rowtot_words() {
    int s = 0;
    for (int j = 1; j < MM; ++j) s += 1 + ho_nk<O>(j);
    return s;
}
template <int O> constexpr int ho_rowtot_off(int j) {
    int s = 0;
    for (int i = 1; i < j; ++i) s += 1 + ho_nk<O>(i);
    return s;
}

template <int C, int MM, int O>
struct WaveHoUndo {
    static constexpr int LQ = MM - 1;
    double qf[LQ][C], qfg[LQ], cpf[LQ][O - 1][C];     // forward accumulators, undone row by row
    double qb[LQ][C], qbg[LQ], scp[LQ][O - 1][C];     // adjoint accumulators (as WaveHoBwd)
    double sv[LQ], sw[LQ][O - 1];                     // adjoint row suffixes for the left neighbour
    double ft[LQ], fr[LQ][O - 1];                     // forward row suffixes (tot_j, row_j[k]) from this lane's first column on

    __device__ __forceinline__ void init(const WaveHoFwd<C, LQ, O>& fw) {
#pragma unroll
        for (int j = 0; j < LQ; ++j) {
            qfg[j] = fw.qg[j];
            qbg[j] = sv[j] = ft[j] = 0.0;
#pragma unroll
            for (int c = 0; c < C; ++c) { qf[j][c] = fw.q[j][c]; qb[j][c] = 0.0; }
#pragma unroll
            for (int k = 0; k < O - 1; ++k) {
                sw[j][k] = fr[j][k] = 0.0;
#pragma unroll
                for (int c = 0; c < C; ++c) { cpf[j][k][c] = fw.cp[j][k][c]; scp[j][k][c] = 0.0; }
            }
        }
    }
    // tT[j], tR[j][k]: totals of lattice row a.  it, ir / iv, iw: the right neighbour's ft, fr / sv, sw of ITS previous step.
    __device__ __forceinline__ void step(const double (&dm)[C], const double (&clev)[MM + 1], const double (&tT)[LQ], const double (&tR)[LQ][O - 1],
                                         const double (&it)[LQ], const double (&ir)[LQ][O - 1], const double (&iv)[LQ], const double (&iw)[LQ][O - 1],
                                         bool first_row, bool first_lane, double (&lam)[C]) {
        double spv[LQ][C], rv[LQ], rw[LQ][O - 1], ut[LQ], ur[LQ][O - 1];
#pragma unroll
        for (int p = 0; p < LQ; ++p) {
            rv[p] = iv[p]; ut[p] = it[p];
#pragma unroll
            for (int k = 0; k < O - 1; ++k) { rw[p][k] = iw[p][k]; ur[p][k] = ir[p][k]; }
#pragma unroll
            for (int c = 0; c < C; ++c) spv[p][c] = c < C - 1 ? qb[p][c < C - 1 ? c + 1 : c] : qbg[p];
        }
#pragma unroll
        for (int c = C - 1; c >= 0; --c) {
            const bool edge = first_row || (first_lane && c == 0);
            double P[LQ], CP[LQ][O - 1], RP[LQ][O - 1], R[LQ][O][O];
            R[0][0][0] = dm[c];
            ho_static_for<1, LQ + 1>([&](auto jc) {
                constexpr int J = decltype(jc)::value, dj = ho_dim<O>(J), nk = ho_nk<O>(J);
                double tot = 0.0;
#pragma unroll
                for (int r = 0; r < dj; ++r)
#pragma unroll
                    for (int k = 0; k < dj; ++k) tot += R[J - 1][r][k];
#pragma unroll
                for (int k = 0; k < O - 1; ++k) {
                    if (k < nk) {
                        double sc = 0.0, sr = 0.0;
#pragma unroll
                        for (int r = 0; r < dj; ++r) { sc += R[J - 1][r][k]; sr += R[J - 1][k][r]; }
                        cpf[J - 1][k][c] -= sc;
                        CP[J - 1][k] = first_row ? 0.0 : cpf[J - 1][k][c];
                        RP[J - 1][k] = (first_lane && c == 0) ? 0.0 : (tR[J - 1][k] - sr) - ur[J - 1][k];
                        ur[J - 1][k] += sr;
                    } else {
                        CP[J - 1][k] = 0.0; RP[J - 1][k] = 0.0;
                    }
                }
                qf[J - 1][c] -= tT[J - 1] - ut[J - 1];                       // Q_J[a-1][b_c]: minus the row's prefix through b_c
                ut[J - 1] += tot;
                const double left = c == 0 ? qfg[J - 1] : qf[J - 1][c > 0 ? c - 1 : 0];
                P[J - 1] = edge ? 0.0 : left - (tT[J - 1] - ut[J - 1]);      // Q_J[a-1][b_c - 1]
                if constexpr (J < LQ) ho_next_grid<O, J>(dm[c], P[J - 1], CP[J - 1], RP[J - 1], R[J - 1], R[J]);
            });
            double l = 0.0, gn[O][O];
#pragma unroll
            for (int r = 0; r < O; ++r)
#pragma unroll
                for (int k = 0; k < O; ++k) gn[r][k] = 0.0;
            ho_static_for_down<MM, 0>([&](auto ic) {
                constexpr int I = decltype(ic)::value, di = ho_dim<O>(I), dn = ho_dim<O>(I + 1);
                double gc[O][O];
#pragma unroll
                for (int r = 0; r < di; ++r)
#pragma unroll
                    for (int k = 0; k < di; ++k) {
                        double g = clev[I];
                        if constexpr (I < MM) {
                            g += spv[I - 1][c];
                            if (k + 1 < dn) g += scp[I - 1][k < O - 1 ? k : 0][c];
                            if (r + 1 < dn) g += rw[I - 1][r < O - 1 ? r : 0];
                            if (r + 1 < dn && k + 1 < dn)
                                g = fma(dm[c] * (1.0 / double((r + 2) * (k + 2))), gn[r + 1 < O ? r + 1 : 0][k + 1 < O ? k + 1 : 0], g);
                        }
                        gc[r][k] = g;
                    }
                if constexpr (I == 1) {
                    l += gc[0][0];
                } else {
                    double s = gc[0][0] * P[I - 2];
#pragma unroll
                    for (int k = 1; k < di; ++k) {
                        s = fma(gc[0][k] * (1.0 / double(k + 1)), CP[I - 2][k - 1], s);
                        s = fma(gc[k][0] * (1.0 / double(k + 1)), RP[I - 2][k - 1], s);
                    }
#pragma unroll
                    for (int r = 1; r < di; ++r)
#pragma unroll
                        for (int k = 1; k < di; ++k) s = fma(gc[r][k] * (1.0 / double((r + 1) * (k + 1))), R[I - 2][r - 1][k - 1], s);
                    l += s;
                }
                if constexpr (I < MM) {
                    rv[I - 1] = fma(dm[c], gn[0][0], rv[I - 1]);
                    qb[I - 1][c] += rv[I - 1];
#pragma unroll
                    for (int k = 0; k + 1 < dn; ++k) {
                        scp[I - 1][k][c] = fma(dm[c] * (1.0 / double(k + 2)), gn[0][k + 1], scp[I - 1][k][c]);
                        rw[I - 1][k] = fma(dm[c] * (1.0 / double(k + 2)), gn[k + 1][0], rw[I - 1][k]);
                    }
                }
#pragma unroll
                for (int r = 0; r < di; ++r)
#pragma unroll
                    for (int k = 0; k < di; ++k) gn[r][k] = gc[r][k];
            });
            lam[c] = l;
        }
#pragma unroll
        for (int p = 0; p < LQ; ++p) {
            qfg[p] -= tT[p] - ut[p];
            qbg[p] += iv[p];
            sv[p] = rv[p]; ft[p] = ut[p];
#pragma unroll
            for (int k = 0; k < O - 1; ++k) { sw[p][k] = rw[p][k]; fr[p][k] = ur[p][k]; }
        }
    }
};

#ifndef HO_UNDO_ATTR
#define HO_UNDO_ATTR
#endif
inline size_t wave_ho_undo_lds(int G, int R1, int order, int M) {
    int w = 0;
    for (int j = 1; j < M; ++j) w += 1 + (((j + 1) < order ? (j + 1) : order) - 1);
    return sizeof(double) * size_t(64 / G) * size_t(R1) * size_t(w);
}

// grid: ngroups / (64 / G) workgroups of one wavefront; dynamic LDS: wave_ho_undo_lds
template <int G, int C, int MM, int O>
__global__ void __launch_bounds__(64) HO_UNDO_ATTR seq_grad_wave_ho_undo_kernel(const WaveHoArgs A) {
    extern __shared__ double ho_rowtot[];
    constexpr int PW = 64 / G, LQ = MM - 1, RW = ho_rowtot_words<O, MM>();
    const int lane = threadIdx.x, lam = lane % G;
    const int grp = blockIdx.x * PW + lane / G;
    const int R1 = A.R1, R2 = A.R2;
    const int TF = R1 + G - 1;
    double* const rt = ho_rowtot + size_t(lane / G) * R1 * RW;
    const int64_t rounds = (A.npairs + A.ngroups - 1) / A.ngroups;
    int nvalid = R2 - C * lam;
    nvalid = nvalid < 0 ? 0 : (nvalid > C ? C : nvalid);

    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t pp = rd * A.ngroups + grp;
        const bool have = pp < A.npairs;
        const int64_t pg = A.pair0 + (have ? pp : 0);
        const int64_t i = A.diag ? pg : pg / A.N2, j = A.diag ? pg : pg % A.N2;
        const double* const dmp = A.dM + size_t(have ? pp : 0) * R1 * R2;
        auto load_dm = [&](int a, double (&dm)[C]) {                   // no branch around the loads: clamped addresses, values selected
            const bool ok = a >= 0 && a < R1;
            const size_t row = size_t(ok ? a : 0) * R2;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int b = C * lam + c;
                const double v = dmp[row + (b < R2 ? b : R2 - 1)];
                dm[c] = (ok && c < nvalid) ? v : 0.0;
            }
        };
        double clev[MM + 1];
#pragma unroll
        for (int p = 0; p <= MM; ++p) clev[p] = (have && p >= 1) ? A.G[p * A.gm + i * A.gi + j * A.gj] : 0.0;

        WaveHoUndo<C, MM, O> bw;
        {
            WaveHoFwd<C, LQ, O> fw;
            fw.reset();
            double dcur[C];
            load_dm(0 - lam, dcur);
            for (int t = 0; t < TF; ++t) {
                double ct[LQ], cr[LQ][O - 1], dnext[C];
#pragma unroll
                for (int m = 0; m < LQ; ++m) {
                    ct[m] = wave_from_left<G>(fw.st[m]);
#pragma unroll
                    for (int k = 0; k < O - 1; ++k) cr[m][k] = wave_from_left<G>(fw.sr[m][k]);
                }
                const int a = t - lam;
                load_dm(a + 1, dnext);
                if (a >= 0 && a < R1) {
                    fw.step(dcur, ct, cr, MM, [](int, int, double) {});
                    if (lam == G - 1) {                                  // the last lane's end-of-chunk prefixes are the row's totals
                        ho_static_for<1, LQ + 1>([&](auto jc) {
                            constexpr int J = decltype(jc)::value, nk = ho_nk<O>(J), off = ho_rowtot_off<O>(J);
                            rt[a * RW + off] = fw.st[J - 1];
#pragma unroll
                            for (int k = 0; k < nk; ++k) rt[a * RW + off + 1 + k] = fw.sr[J - 1][k];
                        });
                    }
                }
#pragma unroll
                for (int c = 0; c < C; ++c) dcur[c] = dnext[c];
            }
            bw.init(fw);
        }
        __syncthreads();                                                 // one wavefront: the row totals are in LDS
        {
            double* const lamrow = A.lam + size_t(have ? pp : 0) * R1 * R2 + C * lam;
            double dcur[C];
            load_dm(R1 - 1 + (G - 1 - lam), dcur);
            for (int u = 0; u < TF; ++u) {
                double it[LQ], ir[LQ][O - 1], iv[LQ], iw[LQ][O - 1], dnext[C];
#pragma unroll
                for (int p = 0; p < LQ; ++p) {
                    it[p] = wave_from_right<G>(bw.ft[p]);
                    iv[p] = wave_from_right<G>(bw.sv[p]);
#pragma unroll
                    for (int k = 0; k < O - 1; ++k) {
                        ir[p][k] = wave_from_right<G>(bw.fr[p][k]);
                        iw[p][k] = wave_from_right<G>(bw.sw[p][k]);
                    }
                }
                const int a = R1 - 1 - (u - (G - 1 - lam));
                load_dm(a - 1, dnext);
                if (a >= 0 && a < R1) {
                    double tT[LQ], tR[LQ][O - 1], lv[C];
                    ho_static_for<1, LQ + 1>([&](auto jc) {
                        constexpr int J = decltype(jc)::value, nk = ho_nk<O>(J), off = ho_rowtot_off<O>(J);
                        tT[J - 1] = rt[a * RW + off];
#pragma unroll
                        for (int k = 0; k < O - 1; ++k) tR[J - 1][k] = k < nk ? rt[a * RW + off + 1 + (k < nk ? k : 0)] : 0.0;
                    });
                    bw.step(dcur, clev, tT, tR, it, ir, iv, iw, a == 0, lam == 0, lv);
                    if (have) {
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            if (c < nvalid) lamrow[size_t(a) * R2 + c] = lv[c];
                    }
                }
#pragma unroll
                for (int c = 0; c < C; ++c) dcur[c] = dnext[c];
            }
        }
        __syncthreads();                                                 // the next pair rewrites the row totals
    }
}

// ---- forward pass: the levels themselves (signature_algs.py:58-71), one sweep per pair.  K_m (m < M) = the 2-D prefix of tot_m at the last cell (the
// group's last lane: columns beyond the lattice pass the row prefixes on unchanged), K_M = the sum over the cells of level M's grid (per-lane sums,
// added up by the group's last lane).  A.lam: the level array, level m of pair (i, j) at [m * gm + i * gi + j * gj].
template <int G, int C, int MM, int O>
__global__ void __launch_bounds__(64) seq_levels_wave_ho_kernel(const WaveHoArgs A) {
    __shared__ double red[64];
    constexpr int PW = 64 / G, LQ = MM - 1;
    const int lane = threadIdx.x, lam = lane % G;
    const int grp = blockIdx.x * PW + lane / G;
    const int R1 = A.R1, R2 = A.R2;
    const int TF = R1 + G - 1;
    const int64_t rounds = (A.npairs + A.ngroups - 1) / A.ngroups;
    int nvalid = R2 - C * lam;
    nvalid = nvalid < 0 ? 0 : (nvalid > C ? C : nvalid);
    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t pp = rd * A.ngroups + grp;
        const bool have = pp < A.npairs;
        const int64_t pg = A.pair0 + (have ? pp : 0);
        const int64_t i = A.diag ? pg : pg / A.N2, j = A.diag ? pg : pg % A.N2;
        const double* const dmp = A.dM + size_t(have ? pp : 0) * R1 * R2;
        auto load_dm = [&](int a, double (&dm)[C]) {
            const bool ok = a >= 0 && a < R1;
            const size_t row = size_t(ok ? a : 0) * R2;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int b = C * lam + c;
                const double v = dmp[row + (b < R2 ? b : R2 - 1)];
                dm[c] = (ok && c < nvalid) ? v : 0.0;
            }
        };
        WaveHoFwd<C, LQ, O> fw;
        fw.reset();
        double ktop = 0.0, dcur[C];
        load_dm(0 - lam, dcur);
        for (int t = 0; t < TF; ++t) {
            double ct[LQ], cr[LQ][O - 1], dnext[C];
#pragma unroll
            for (int m = 0; m < LQ; ++m) {
                ct[m] = wave_from_left<G>(fw.st[m]);
#pragma unroll
                for (int k = 0; k < O - 1; ++k) cr[m][k] = wave_from_left<G>(fw.sr[m][k]);
            }
            const int a = t - lam;
            load_dm(a + 1, dnext);
            if (a >= 0 && a < R1) fw.template step_impl<true>(dcur, ct, cr, MM, [](int, int, double) {}, ktop);
#pragma unroll
            for (int c = 0; c < C; ++c) dcur[c] = dnext[c];
        }
        __syncthreads();
        red[lane] = ktop;
        __syncthreads();
        if (lam == G - 1 && have) {
            double s = 0.0;
            for (int l = 0; l < G; ++l) s += red[lane - (G - 1) + l];
            double* const o = A.lam + i * A.gi + j * A.gj;
            o[0] = 1.0;                                                  // signature_algs.py:49-53
#pragma unroll
            for (int m = 1; m <= LQ; ++m) o[int64_t(m) * A.gm] = fw.q[m - 1][C - 1];
            o[int64_t(MM) * A.gm] = s;
        }
    }
}

// ---- first order from a dM lattice in memory (round 6): the scratch-free sweeps of seq_lam_undo_kernel (grad_wave_kernel.hpp: WaveFwd + WaveUndo) with the lane shapes
// above and dM read instead of evaluated -- for the wide route's MANY SHORT lattices (a Gram of sequences of <= 64 observations at 17+ columns), which its own lattice
// kernels sweep one per 64-lane wavefront however short they are.  Dynamic LDS: (64 / G) * R1 * LQ doubles (the row totals).
template <int G, int C, int LQ>
__global__ void __launch_bounds__(64) seq_grad_wave_o1_kernel(const WaveHoArgs A) {
    extern __shared__ double o1_rowtot[];
    constexpr int PW = 64 / G;
    const int lane = threadIdx.x, lam = lane % G;
    const int grp = blockIdx.x * PW + lane / G;
    const int R1 = A.R1, R2 = A.R2, M = A.M;
    const int TF = R1 + G - 1;
    double* const rt = o1_rowtot + size_t(lane / G) * R1 * LQ;
    const int64_t rounds = (A.npairs + A.ngroups - 1) / A.ngroups;
    int nvalid = R2 - C * lam;
    nvalid = nvalid < 0 ? 0 : (nvalid > C ? C : nvalid);
    const int last_lane = R2 > 0 ? (R2 - 1) / C : 0;
    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t pp = rd * A.ngroups + grp;
        const bool have = pp < A.npairs;
        const int64_t pg = A.pair0 + (have ? pp : 0);
        const int64_t i = A.diag ? pg : pg / A.N2, j = A.diag ? pg : pg % A.N2;
        const double* const dmp = A.dM + size_t(have ? pp : 0) * R1 * R2;
        auto load_dm = [&](int a, double (&dm)[C]) {
            const bool ok = a >= 0 && a < R1;
            const size_t row = size_t(ok ? a : 0) * R2;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int b = C * lam + c;
                const double v = dmp[row + (b < R2 ? b : R2 - 1)];
                dm[c] = (ok && c < nvalid) ? v : 0.0;
            }
        };
        double clev[LQ + 2];
#pragma unroll
        for (int p = 0; p < LQ + 2; ++p) clev[p] = (have && p >= 1 && p <= M) ? A.G[p * A.gm + i * A.gi + j * A.gj] : 0.0;
        WaveFwd<C, LQ> fw;
        fw.reset();
        double dcur[C];
        load_dm(0 - lam, dcur);
        for (int t = 0; t < TF; ++t) {
            double cin[LQ + 2], dnext[C];
            cin[0] = 0.0;
#pragma unroll
            for (int m = 1; m < LQ + 2; ++m) cin[m] = wave_from_left<G>(fw.sout[m]);
            const int a = t - lam;
            load_dm(a + 1, dnext);
            if (a >= 0 && a < R1) {
                fw.step(dcur, cin, M);
                if (lam == last_lane) {
#pragma unroll
                    for (int m = 1; m <= LQ; ++m) rt[a * LQ + m - 1] = m < M ? fw.sout[m] : 0.0;
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) dcur[c] = dnext[c];
        }
        __syncthreads();
        WaveUndo<C, LQ> bw;
        bw.init(fw);
        double* const lamrow = A.lam + size_t(have ? pp : 0) * R1 * R2 + C * lam;
        load_dm(R1 - 1 + (G - 1 - lam), dcur);
        for (int u = 0; u < TF; ++u) {
            double sufin[LQ], svin[LQ], dnext[C];
#pragma unroll
            for (int p = 0; p < LQ; ++p) {
                sufin[p] = wave_from_right<G>(bw.sufout[p]);
                svin[p] = wave_from_right<G>(bw.svout[p]);
            }
            const int a = R1 - 1 - (u - (G - 1 - lam));
            load_dm(a - 1, dnext);
            if (a >= 0 && a < R1) {
                double rtv[LQ], lv[C];
#pragma unroll
                for (int p = 0; p < LQ; ++p) rtv[p] = rt[a * LQ + p];
                bw.step(dcur, clev, rtv, sufin, svin, M, a == 0, lam == 0, lv);
                if (have) {
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        if (c < nvalid) lamrow[size_t(a) * R2 + c] = lv[c];
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) dcur[c] = dnext[c];
        }
        __syncthreads();
    }
}

}  // namespace gpsig
