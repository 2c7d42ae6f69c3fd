// CPU lock-step emulator of the gfx950 seq-gram kernel -- TEST INFRASTRUCTURE ONLY.
//
// It runs the very same per-lane code (gpsig_amd/csrc/seq_core.hpp: seq_step, LaneCtl; seq_args.hpp:
// seq_emit, seq_build_tasks, seq_ring_depth) for the 64 lanes of a wavefront in lock step, with the
// DPP shifts replaced by a snapshot of the left neighbour's registers and the LDS ring replaced by a
// plain array that is refilled at exactly the steps the kernel refills it.  The CPU test-suite uses it
// to check the skewed-lane recursion, pair hand-over, ring-slot reuse, task coverage and the epilogue
// against the oracle without a GPU.  The product never links or loads this file.
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>

#include "seq_args.hpp"
#include "seq_core.hpp"

using namespace gpsig;

template <typename T, int G, int C, int D, int MMAX, int MODE, bool EXACT, int OMAX = 0>
static void emu_task(const SeqGramArgs& A, const SeqTask& tk) {
    using Lane = typename std::conditional<OMAX == 0, SeqLane<T, C, D, MMAX, MODE>, SeqLaneHO<T, C, D, MMAX, (OMAX > 0 ? OMAX : 1), MODE>>::type;
    const int M = EXACT ? MMAX : A.M;
    const int R1 = A.R1, RS = A.RS, nslot = A.nslot, nx = tk.nx;
    const T* xrec = static_cast<const T*>(A.xrec);
    const T* yrec = static_cast<const T*>(A.yrec);
    // [zero row][ring]; the ring is NaN-poisoned: reads of slots that were never filled must not matter
    std::vector<T> lds(size_t(RS) + size_t(nslot) * A.slot_elems, T(0) / T(0));
    for (int k = 0; k < RS; ++k) lds[k] = T(0);
    T* const ring_base = lds.data() + RS;
    Lane L[64];
    LaneCtl ctl[64];
    int64_t jj[64]; bool jvalid[64]; int rlo[64], rhi[64];
    for (int lane = 0; lane < 64; ++lane) {
        const int lam = lane & (G - 1), grp = lane / G;
        jj[lane] = int64_t(tk.y0) + grp;
        jvalid[lane] = jj[lane] < A.N2;
        L[lane].init();
        for (int r = 0; r < C; ++r) {
            const int row = C * lam + r;
            const bool ok = jvalid[lane] && row < A.R2;
            T ys = 0;
            for (int f = 0; f < D; ++f) {
                T v = ok ? yrec[jj[lane] * A.yrec_stride + int64_t(row) * RS + f] : T(0);
                L[lane].y[r][f] = v;
                ys = std::fma(v, v, ys);
            }
            L[lane].y2[r] = ys;
        }
        rlo[lane] = lam == 0 ? 1 : 0;
        int h = A.R2 - C * lam;
        rhi[lane] = h < 0 ? 0 : (h > C ? C : h);
        ctl[lane].init(lam, RS);
    }
    auto stage = [&](int p, int slot) {
        int64_t i = int64_t(tk.x0) + p;
        if (i >= A.N1) i -= A.N1;
        std::memcpy(ring_base + size_t(slot) * A.slot_elems, xrec + i * A.xrec_stride, sizeof(T) * A.slot_elems);
    };
    stage(0, 0);
    const int nsteps = nx * R1 + G;
    const int ring_elems = nslot * A.slot_elems;
    int a_u = 0, k_u = 0, slot_next = 1 % nslot;
    T* out = static_cast<T*>(A.out);
    for (int t = 0; t < nsteps; ++t) {
        if (a_u == A.issue_at && k_u + 1 < nx) { stage(k_u + 1, slot_next); if (++slot_next == nslot) slot_next = 0; }
        if (++a_u == R1) { a_u = 0; ++k_u; }
        // Order of events inside one step of the real wave: (1) the pair-boundary block (emit + reset) runs for
        // the lanes that sit on a boundary, (2) every lane reads its left neighbour's hand-over registers (DPP
        // shifts issued inside seq_step, before those registers are rewritten), (3) arithmetic.  The snapshot
        // therefore has to be taken AFTER the boundary blocks of all lanes.
        for (int lane = 0; lane < 64; ++lane) {
            const int lam = lane & (G - 1);
            if (ctl[lane].begin_step(nx, R1, RS, A.slot_elems, ring_elems)) {
                if (lam == G - 1 && ctl[lane].p >= 1 && jvalid[lane]) {
                    int64_t i = int64_t(tk.x0) + (ctl[lane].p - 1);
                    if (i >= A.N1) i -= A.N1;
                    seq_emit<T>(L[lane], A, i, jj[lane], M, [&](int64_t off, T v) { out[off] = v; });
                }
                L[lane].reset();
            }
        }
        NbrSnapshot<T, MMAX> snap[64];
        for (int lane = 0; lane < 64; ++lane) {
            const int lam = lane & (G - 1);
            NbrSnapshot<T, MMAX>& S = snap[lane];
            for (int m = 0; m < MMAX; ++m) S.s[m] = lam ? L[lane - 1].s[m] : T(0);
            S.klast = lam ? L[lane - 1].klast : T(0);
            if constexpr (Lane::HIGHER_ORDER) {
                for (int m = 0; m < Lane::NQ; ++m)
                    for (int r = 0; r < Lane::NO; ++r) S.w[m][r] = lam ? L[lane - 1].w[m][r] : T(0);
            }
        }
        for (int lane = 0; lane < 64; ++lane) {
            const T* rowp = lds.data() + ctl[lane].rowoff;
            T xr[D];
            for (int f = 0; f < D; ++f) xr[f] = rowp[f];
            const bool dummy = ctl[lane].row0;
            seq_step(L[lane], snap[lane], xr, M, A.order, dummy, rlo[lane], rhi[lane], A.kind, T(A.p0), T(A.p1));
            ctl[lane].end_step();
        }
    }
}

template <typename T, int G, int C, int D, int MMAX, int MODE, bool EXACT, int OMAX = 0>
static void emu_run(const SeqGramArgs& A, int ntasks) {
    for (int b = 0; b < ntasks; ++b) emu_task<T, G, C, D, MMAX, MODE, EXACT, OMAX>(A, A.tasks[b]);
}

#define TRY(G_, C_, D_, MM_, EX_)                                                                         \
    if (G == G_ && C == C_ && D == D_ && MMAX == MM_ && exact == int(EX_)) {                              \
        if (mode == MODE_INC) emu_run<double, G_, C_, D_, MM_, MODE_INC, EX_>(*A, ntasks);                 \
        else if (mode == MODE_PT_DIFF) emu_run<double, G_, C_, D_, MM_, MODE_PT_DIFF, EX_>(*A, ntasks);    \
        else emu_run<double, G_, C_, D_, MM_, MODE_PT_NODIFF, EX_>(*A, ntasks);                            \
        return 0;                                                                                         \
    }

#define TRY_HO(G_, C_, D_, MM_, OM_)                                                                             \
    if (G == G_ && C == C_ && D == D_ && MMAX == MM_ && OMAX == OM_) {                                           \
        if (mode == MODE_INC) emu_run<double, G_, C_, D_, MM_, MODE_INC, false, OM_>(*A, ntasks);                \
        else if (mode == MODE_PT_DIFF) emu_run<double, G_, C_, D_, MM_, MODE_PT_DIFF, false, OM_>(*A, ntasks);   \
        else return -2;                                                                                         \
        return 0;                                                                                               \
    }

extern "C" {

// higher-order kernels: emulator table {G, C, D, MMAX, OMAX}
int emu_seq_gram_ho(int G, int C, int D, int MMAX, int OMAX, int mode, const SeqGramArgs* A, int ntasks) {
    TRY_HO(64, 1, 4, 6, 6)
    TRY_HO(64, 2, 4, 5, 3)
    TRY_HO(16, 2, 4, 4, 4)
    return -1;
}

int emu_seq_gram(int G, int C, int D, int MMAX, int mode, int exact, const SeqGramArgs* A, int ntasks) {
    TRY(16, 4, 8, 5, true)
    TRY(16, 4, 8, 4, true)
    TRY(16, 2, 4, 4, true)
    TRY(16, 2, 4, 5, true)
    TRY(16, 1, 2, 3, true)
    TRY(16, 1, 4, 8, false)
    TRY(16, 2, 4, 8, false)
    TRY(16, 4, 4, 8, false)
    TRY(16, 8, 2, 8, false)
    TRY(64, 1, 4, 8, false)
    TRY(64, 2, 2, 4, true)
    return -1;
}

int emu_ring_depth(int G, int R1) { return seq_ring_depth(G, R1); }
int emu_ring_issue_at(int G, int R1) { return seq_ring(G, R1).issue_at; }

// returns the number of tasks; writes at most cap of them
int emu_build_tasks(int64_t N1, int64_t N2, int ypb, int pred, int max_run, int shard_index, int shard_count,
                    int64_t y_begin, int64_t y_end, SeqTask* out, int cap) {
    std::vector<SeqTask> t = seq_build_tasks(N1, N2, ypb, pred, max_run, shard_index, shard_count, y_begin, y_end);
    for (size_t k = 0; k < t.size() && int(k) < cap; ++k) out[k] = t[k];
    return int(t.size());
}

}  // extern "C"

#include "seq_configs.hpp"

// The emulator's own (small) config table, searched with the product's selection function.
static const SeqConfig EMU_TABLE[] = {
    {16, 4, 8, 5, true}, {16, 4, 8, 4, true}, {16, 2, 4, 4, true}, {16, 2, 4, 5, true}, {16, 1, 2, 3, true},
    {16, 1, 4, 8, false}, {16, 2, 4, 8, false}, {16, 4, 4, 8, false}, {16, 8, 2, 8, false},
    {64, 1, 4, 8, false}, {64, 2, 2, 4, true},
};

extern "C" {
int emu_select(int Ry, int d, int M, int allow_exact, int* cfg /*G,C,D,MMAX,exact*/) {
    const int n = int(sizeof(EMU_TABLE) / sizeof(EMU_TABLE[0]));
    int k = seq_select(EMU_TABLE, n, Ry, d, M, allow_exact != 0);
    if (k < 0) return -1;
    cfg[0] = EMU_TABLE[k].G; cfg[1] = EMU_TABLE[k].C; cfg[2] = EMU_TABLE[k].D; cfg[3] = EMU_TABLE[k].MMAX;
    cfg[4] = EMU_TABLE[k].exact;
    return k;
}
int emu_geometry(int base_kind, int difference, int L, int D, int elem_bytes, int* out /*mode,rows,RS,rec_elems*/) {
    SeqGeom g = seq_geometry(base_kind, difference, L, D, elem_bytes);
    out[0] = g.mode; out[1] = g.rows; out[2] = g.RS; out[3] = g.rec_elems;
    return 0;
}
}
