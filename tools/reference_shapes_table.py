"""profiles/r06_reference_shapes.txt from the GPU visits' records (tools/gpu_r6_a.sh: round 5's routes; tools/gpu_r6_h.sh: this round's, with one kernel
trace per data set):  python tools/reference_shapes_table.py gpurun_out/r06a/reference_shapes_before.jsonl gpurun_out/r06h > profiles/r06_reference_shapes.txt"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import reference_shapes as RS  # noqa: E402


def load(path, route):
    out = {}
    for line in open(path):
        line = line.strip()
        if line.startswith("{"):
            r = json.loads(line)
            if r.get("route") == route and "error" not in r:
                out[r["dataset"]] = r
    return out


def kernels_of(path, top=7):
    """the library's kernels of one data set's trace, by total time: name (calls, average us)"""
    rows = []
    if not os.path.exists(path):
        return rows
    for line in open(path):
        m = re.match(r"^(?:void )?(gpsig::\S+|Cijk_\S+|void rocsolver\S+|rocblas\S+)", line)
        if not m:
            continue
        f = line[112:].split()
        if len(f) < 9:
            continue
        name = m.group(1).replace("gpsig::", "")
        name = re.sub(r"^Cijk_(A\w+?_B\w+?)_.*", r"rocBLAS dgemm (\1)", name)
        rows.append((float(f[6]), name[:58], int(f[5]), float(f[7])))
    rows.sort(reverse=True)
    return rows[:top]


def main():
    before = load(sys.argv[1], "auto")
    d = sys.argv[2]
    after = load(os.path.join(d, "shapes_clean.jsonl"), "auto")
    names = sorted(RS.DATASETS, key=lambda n: (RS.shape_of(n)["d_eff"], RS.shape_of(n)["L"]))
    print("The reference's own run settings as shapes (benchmarks/run_gpsig_benchmarks.py:32 on benchmarks/datasets.json; tools/reference_shapes.py):")
    print("num_levels=4, 500 inducing tensors with increments, num_lags=1 on time-augmented data (2 (n_features + 1) columns), minibatch 50, SignatureRBF,")
    print("float64, one MI355X; synthetic paths of each data set's shape.  Times in ms: forward / forward + backward of each covariance (autodiff module,")
    print("scaled inputs), of the three together from the raw inputs (covs), and one whole SVGP step (-ELBO forward + backward: + conditional, KL, likelihood).")
    print("round 5 = the library as round 5 left it (auto route of tools/gpu_r6_a.sh); round 6 = now.\n")
    hdr = "%-22s %5s %4s %3s | %-21s | %-15s %-15s %-15s | %-15s %9s" % ("data set", "cols", "L", "N", "round 5: covs f+b / step", "Kzz f / f+b", "Kzx f / f+b",
                                                                       "Kxx-diag f / f+b", "covs f / f+b", "step")
    print(hdr)
    print("-" * len(hdr))
    for n in names:
        s = RS.shape_of(n)
        b, a = before.get(n), after.get(n)
        bs = "%9.2f / %8.2f" % (b["covs_fwd_bwd_ms"], b["step_ms"]) if b else "       (not run)    "
        if a:
            f2 = lambda k: "%6.2f / %6.2f" % (a[k + "_fwd_ms"], a[k + "_fwd_bwd_ms"])     # noqa: E731
            print("%-22s %5d %4d %3d | %-21s | %-15s %-15s %-15s | %-15s %9.2f" % (n, s["d_eff"], s["L"], s["N"], bs, f2("kzz"), f2("kzx"), f2("kxx_diag"),
                                                                                 "%6.2f / %6.2f" % (a["covs_fwd_ms"], a["covs_fwd_bwd_ms"]), a["step_ms"]))
        else:
            print("%-22s %5d %4d %3d | %-21s | (no record)" % (n, s["d_eff"], s["L"], s["N"], bs))
    print("\n(KickvsPunch, Shapes were not in round 5's visit: same columns / lengths as WalkvsRun, DigitShapes.)")
    print("\nWhich kernels ran (rocprofv3 --kernel-trace of tools/reference_shapes.py <data set>: forward, forward + backward of each covariance, of the three, and")
    print("the step, 3-5 repetitions each; total us | calls | average us).  No seq_levels_generic*, tens_vs_seq_kernel or tens_gram_kernel anywhere:")
    for n in names:
        s = RS.shape_of(n)
        print("\n%s (%d columns, L = %d)" % (n, s["d_eff"], s["L"]))
        path = os.path.join(d, "kernels_%s.txt" % n)
        for tot, name, calls, avg in kernels_of(path):
            print("    %-60s %10.0f %5d %9.1f" % (name, tot, calls, avg))
        if os.path.exists(path):
            txt = open(path).read()
            bad = [k for k in ("seq_levels_generic", "tens_vs_seq_kernel<", "tens_gram_kernel<", "seq_lam_undo") if k in txt]
            print("    older mappings / fallbacks in the trace: %s" % (", ".join(bad) if bad else "none"))


if __name__ == "__main__":
    main()
