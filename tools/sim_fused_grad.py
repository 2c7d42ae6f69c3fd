"""Lock-step model of grad_fused_kernel.hpp (round 5): the two-role reverse pass of one sequence pair, in numpy.

One pair group = G lanes of C lattice columns.  Two wavefronts work on it: the EVALUATOR (base-kernel values of a row of the
streamed sequence against the lane's points, the double increments dm, and -- in the backward sweep -- the contraction of
Lam with the kernel's derivative for both sides) and the SWEEPER (forward recursion, then the recursion undone row by row,
Lam out).  They exchange dm / Lam through same-lane slots, one barrier per interval.  This script replays the intervals with
arrays over lanes, DPP shifts as shifted copies, and checks the gradients against torch autograd of the plain recursion.

Run: python tools/sim_fused_grad.py
"""
import os

import numpy as np
import torch

DIFF = os.environ.get("SIM_DIFF", "1") != "0"      # 0: difference=False (the lattice is the kernel matrix of the points)
G, C = int(os.environ.get("SIM_G", "16")), int(os.environ.get("SIM_C", "4"))    # lanes per pair group (16: four pairs per wavefront on the GPU, 64: one); columns per lane


def from_left(v):      # lane l <- lane l-1, 0 into lane 0
    out = np.zeros_like(v)
    out[1:] = v[:-1]
    return out


def from_right(v):     # lane l <- lane l+1, 0 into the last lane
    out = np.zeros_like(v)
    out[:-1] = v[1:]
    return out


def reference(x, y, clev, diff=True):
    """levels of the first-order signature kernel with the RBF base kernel on points, and d sum_m clev[m] K_m / d(x, y)"""
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yt = torch.tensor(y, dtype=torch.float64, requires_grad=True)
    d2 = (xt * xt).sum(1)[:, None] + (yt * yt).sum(1)[None, :] - 2 * xt @ yt.T
    k = torch.exp(-d2 / 2)
    dm = k[1:, 1:] - k[1:, :-1] - k[:-1, 1:] + k[:-1, :-1] if diff else k
    M = len(clev) - 1
    Rm = dm
    loss = clev[1] * Rm.sum()
    for m in range(2, M + 1):
        Q = Rm.cumsum(0).cumsum(1)
        Qs = torch.zeros_like(Q)
        Qs[1:, 1:] = Q[:-1, :-1]
        Rm = dm * Qs
        loss = loss + clev[m] * Rm.sum()
    loss.backward()
    return xt.grad.numpy(), yt.grad.numpy()


def fused(x, y, clev, diff=True):
    L1, D = x.shape
    L2 = y.shape[0]
    R1, R2 = (L1 - 1, L2 - 1) if diff else (L1, L2)      # difference=False: the lattice is the kernel matrix itself, no increments along either side
    M = len(clev) - 1
    LQ = M - 1
    TF = R1 + G - 1
    ln = np.arange(G)
    b0 = C * ln
    nvalid = np.clip(R2 - b0, 0, C)
    last_lane = (R2 - 1) // C
    cs = np.sqrt(np.log2(np.e))
    xs = x * cs                                  # staged record: prescaled rows and -|x'|^2 / 2
    hx = -0.5 * (xs * xs).sum(1)
    # evaluator: the lane's C + 1 points (beyond the sequence the last point repeats: dm == 0 there without a mask)
    yp = np.zeros((G, C + 1, D))
    for c in range(C + 1):
        yp[:, c] = y[np.minimum(b0 + c, L2 - 1)] * cs
    hy = -0.5 * (yp * yp).sum(2)

    def kappa_row(p):                            # p: per-lane row index (clamped), -> (G, C+1)
        pc = np.clip(p, 0, L1 - 1)
        return np.exp2(np.einsum('lf,lcf->lc', xs[pc], yp) + hx[pc][:, None] + hy)

    def mask_cols(v):
        out = v.copy()
        for c in range(C):
            out[nvalid <= c, c] = 0.0
        return out

    # ---------------- forward sweep, evaluator one interval ahead of the sweeper.  Here a lane's four columns are b0-1 .. b0+2 -- the
    # differences that END at its own points b0 .. b0+3, the kernel value at b0-1 being the left neighbour's last, one interval old (it
    # runs one row ahead): four evaluations per row, as in the evaluation kernel (seq_core.hpp).  Lane 0's column -1 is no column: dm == 0.
    dmslot = np.zeros((2, G, C))
    q = np.zeros((G, LQ, C)); qg = np.zeros((G, LQ)); sout = np.zeros((G, LQ + 2))
    rowtot = np.zeros((R1, LQ))

    def col_diffs(k, kleft):
        kl = kleft.copy()
        kl[0] = k[0, 0]
        nd = np.empty((G, C))
        nd[:, 0] = k[:, 0] - kl
        nd[:, 1:] = k[:, 1:C] - k[:, 0:C - 1]
        return nd

    k = kappa_row(np.zeros(G, int))
    rd = col_diffs(k, from_left(k[:, C - 1]))
    k3 = k[:, C - 1].copy()

    def mask_pts(v):                              # difference=False: columns are points; beyond the sequence dm must be forced to zero
        out = v.copy()
        for c in range(C):
            out[nvalid <= c, c] = 0.0
        return out
    for tau in range(TF + 1):
        if tau < TF:
            a = tau - ln
            if diff:
                k = kappa_row(a + 1)              # (clamped: a lane ahead of its rows evaluates row 0 again, so rd needs no guard)
                nd = col_diffs(k, from_left(k3))
                k3 = k[:, C - 1].copy()
                dmslot_new = nd - rd
                rd = nd
            else:
                dmslot_new = mask_pts(kappa_row(a)[:, :C])
        if tau >= 1:
            t = tau - 1
            a = t - ln
            act = (a >= 0) & (a < R1)
            dm_in = dmslot[t % 2]
            cin = np.zeros((G, LQ + 2))
            for m in range(1, LQ + 2):
                cin[:, m] = from_left(sout[:, m])
            qn, qgn, soutn = q.copy(), qg.copy(), sout.copy()
            for m in range(LQ + 1, 1, -1):
                if m <= M:
                    lo = m - 2
                    s_ = cin[:, m].copy()
                    for c in range(C):
                        s_ = s_ + dm_in[:, c] * (qg[:, lo] if c == 0 else q[:, lo, c - 1])
                        if m < M:
                            qn[:, m - 1, c] = q[:, m - 1, c] + s_
                    soutn[:, m] = s_
                    if m < M:
                        qgn[:, m - 1] = qg[:, m - 1] + cin[:, m]
            s_ = cin[:, 1].copy()
            for c in range(C):
                s_ = s_ + dm_in[:, c]
                if 1 < M:
                    qn[:, 0, c] = q[:, 0, c] + s_
            soutn[:, 1] = s_
            if 1 < M:
                qgn[:, 0] = qg[:, 0] + cin[:, 1]
            q = np.where(act[:, None, None], qn, q); qg = np.where(act[:, None], qgn, qg); sout = np.where(act[:, None], soutn, sout)
            if act[G - 1]:                        # dm == 0 beyond the sequence: the last lane's prefix is the row total whatever R2
                for m in range(1, LQ + 1):
                    rowtot[a[G - 1], m - 1] = sout[G - 1, m]
        if tau < TF:
            dmslot[tau % 2] = dmslot_new

    # ---------------- backward sweep.  Interval i: the evaluator's kernel row a = R1 + (G-1-ln) - i (points b0 .. b0+3; the value at
    # b0+4 is the right neighbour's first, one interval old); the sweeper's step i - 2 and the adjoint W = -H * kappa of point row a + 2
    # with the kernel values of interval i - 3; the evaluator's contraction of the W handed over in interval i - 1.
    # the backward sweep's columns are b0 .. b0+3: one to the right of the forward sweep's
    if diff:
        qf = np.zeros((G, LQ, C)); qfg = q[:, :, 0].copy()
        qf[:, :, :C - 1] = q[:, :, 1:]
        for m in range(LQ):
            qf[:, m, C - 1] = from_right(q[:, m, 0])
    else:                                         # difference=False: columns are the lane's points in both sweeps
        qf, qfg = q.copy(), qg.copy()
    # the upstream gradients ride in the suffix sums from the start (U_p = c_p + Qb_p: one add per cell less)
    qb = np.zeros((G, LQ, C)); qbg = np.zeros((G, LQ)); svout = np.zeros((G, LQ)); sufout = np.zeros((G, LQ))
    for p_ in range(1, M):
        qb[:, p_ - 1, :] = clev[p_]
        qbg[:, p_ - 1] = clev[p_]
    wslot = np.full((2, G, C), np.nan)
    KH = 5
    khist = np.zeros((KH, G, C))                  # same-lane ring keyed by the interval (cleared once per task: finite whatever is read)
    k0 = np.zeros(G)
    lamk = np.zeros((G, C)); Eprev = np.zeros((G, C)); Pout = np.zeros((G, D + 1))
    Ay = np.zeros((G, C, D)); By = np.zeros((G, C))
    gxa = np.zeros((L1, D + 1))
    for i in range(TF + 5):
        # ---- evaluator, evaluation
        if i <= TF:
            a = R1 + (G - 1 - ln) - i
            k = kappa_row(a)
            if diff:
                k[:, C] = from_right(k0)
                k[G - 1, C] = k[G - 1, C - 1]       # the last lane has no neighbour: its last column (63) is never a lattice column, dm == 0 there
                k0 = k[:, 0].copy()
                nd = k[:, 1:] - k[:, :-1]
                dm_new = rd - nd
                rd = nd
            else:
                dm_new = mask_pts(k[:, :C])
            khist[i % KH] = k[:, :C]
        # ---- evaluator, contraction of the W of interval i - 1
        if i >= 3:
            p = R1 + (4 if diff else 2) + (G - 1 - ln) - i
            W = wslot[(i - 1) % 2]
            assert np.isfinite(W).all()
            pc = np.clip(p, 0, L1 - 1)
            Ay += W[:, :, None] * xs[pc][:, None, :]
            By += W
            Pin = from_right(Pout)
            Pout = Pin.copy()
            Pout[:, :D] += np.einsum('lc,lcf->lf', W, yp[:, :C])
            Pout[:, D] += W.sum(1)
            if 0 <= p[0] <= L1 - 1:
                gxa[p[0]] += Pout[0]
        # ---- sweeper: step i - 2, then E / H / W
        if 2 <= i <= TF + 3:
            a = R1 - 1 - ((i - 2) - (G - 1 - ln))
            act = (a >= 0) & (a < R1)
            dm_in = dmslot[(i - 1) % 2]
            sufin = np.stack([from_right(sufout[:, p_]) for p_ in range(LQ)], 1)
            svin = np.stack([from_right(svout[:, p_]) for p_ in range(LQ)], 1)
            rt = rowtot[np.clip(a, 0, R1 - 1)]
            first_row = a == 0
            first_lane = ln == 0
            qfn, qfgn, sufn = qf.copy(), qfg.copy(), sufout.copy()
            Dm = np.ones((G, LQ + 1, C))
            for m in range(1, LQ + 1):
                if m < M:
                    vv = sufin[:, m - 1] - rt[:, m - 1]
                    for c in range(C - 1, -1, -1):
                        qfn[:, m - 1, c] = qf[:, m - 1, c] + vv
                        vv = vv + dm_in[:, c] * Dm[:, m - 1, c]
                    qfgn[:, m - 1] = qfg[:, m - 1] + vv
                    sufn[:, m - 1] = vv + rt[:, m - 1]
                for c in range(C):
                    dd = qfgn[:, m - 1] if c == 0 else qfn[:, m - 1, c - 1]
                    dd = np.where((first_lane & (c == 0)) | (m >= M), 0.0, dd)      # (row 0 is not forced to zero: its residue is any row's)
                    Dm[:, m, c] = dd
            U = np.zeros((G, LQ + 2, C))
            for p_ in range(1, LQ + 2):
                for c in range(C):
                    pi = min(p_ - 1, LQ - 1)
                    if p_ < M:
                        U[:, p_, c] = qb[:, pi, c + 1] if c < C - 1 else qbg[:, pi]
                    else:
                        U[:, p_, c] = clev[p_] if p_ == M else 0.0
            lam = np.zeros((G, C))
            for c in range(C):
                l_ = U[:, 1, c].copy()
                for p_ in range(2, LQ + 2):
                    if p_ <= M:
                        l_ = l_ + Dm[:, p_ - 1, c] * U[:, p_, c]
                lam[:, c] = l_
            qbn, qbgn, svn = qb.copy(), qbg.copy(), svout.copy()
            for p_ in range(1, LQ + 1):
                if p_ < M:
                    sv = svin[:, p_ - 1].copy()
                    for c in range(C - 1, -1, -1):
                        sv = sv + dm_in[:, c] * U[:, p_ + 1, c]
                        qbn[:, p_ - 1, c] = qb[:, p_ - 1, c] + sv
                    svn[:, p_ - 1] = sv
                    qbgn[:, p_ - 1] = qbg[:, p_ - 1] + svin[:, p_ - 1]
            A3, A2 = act[:, None, None], act[:, None]
            qf = np.where(A3, qfn, qf); qfg = np.where(A2, qfgn, qfg); sufout = np.where(A2, sufn, sufout)
            qb = np.where(A3, qbn, qb); qbg = np.where(A2, qbgn, qbg); svout = np.where(A2, svn, svout)
            li = np.where(A2, mask_cols(lam), 0.0)
            if not diff:                          # H = Lam: the kernel values of this row were evaluated in the previous interval
                w_new = -li * khist[(i - 1) % KH]
            Enew = li - lamk
            lamk = li
            Eleft = from_left(Enew[:, C - 1])
            H = np.zeros((G, C))
            H[:, 0] = Eleft - Eprev[:, 0]
            for c in range(1, C):
                H[:, c] = Eprev[:, c - 1] - Eprev[:, c]
            Eprev = Enew
            if diff:
                w_new = -H * khist[(i - 3) % KH]   # H == 0 outside the point rows 0 .. R1 by construction
        if i <= TF:
            dmslot[i % 2] = dm_new
        if 2 <= i <= TF + 3:
            wslot[i % 2] = w_new
    # flush
    gx = (gxa[:, D:D + 1] * xs - gxa[:, :D]) / cs
    gy = np.zeros_like(y)
    for l in range(G):
        for c in range(C):
            qq = b0[l] + c
            if qq < L2:
                gy[qq] += (By[l, c] * yp[l, c] - Ay[l, c]) / cs
    return gx, gy


def main():
    rng = np.random.default_rng(5)
    worst = 0.0
    shapes64 = ((70, 200, 4, 3), (130, 256, 2, 5), (5, 65, 3, 2))
    shapes_c2 = ((40, 32, 12, 4), (9, 17, 16, 3), (3, 2, 9, 2))
    for (L1, L2, D, M) in ((100, 128, 4, 3), (70, 65, 8, 5)) if G == 32 else shapes64 if G == 64 else shapes_c2 if C == 2 else ((64, 64, 8, 5), (9, 64, 8, 4), (64, 33, 4, 3), (5, 7, 3, 2), (20, 62, 8, 5), (3, 2, 2, 3), (2, 64, 8, 5), (64, 64, 1, 2)):
        x = np.cumsum(rng.standard_normal((L1, D)) * 0.3, 0)
        y = np.cumsum(rng.standard_normal((L2, D)) * 0.3, 0)
        clev = np.concatenate([[0.0], rng.standard_normal(M)])
        gx0, gy0 = reference(x, y, clev, DIFF)
        gx1, gy1 = fused(x, y, clev, DIFF)
        ex = np.abs(gx1 - gx0).max() / max(np.abs(gx0).max(), 1e-300)
        ey = np.abs(gy1 - gy0).max() / max(np.abs(gy0).max(), 1e-300)
        print(f"L1={L1} L2={L2} d={D} M={M}: rel err x {ex:.2e}  y {ey:.2e}")
        worst = max(worst, ex, ey)
    print("worst", worst)
    assert worst < 1e-9


if __name__ == "__main__":
    main()
