#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite) results: per-kernel stats and per-dispatch PMC values.
    python tools/rocprof_summary.py stats <results.db>
    python tools/rocprof_summary.py pmc <results.db> [kernel-name-substring]
"""
import sqlite3
import sys
from collections import defaultdict


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, grid_x, workgroup_x, duration, vgpr_count, sgpr_count, lds_size from kernels"))
    agg = defaultdict(list)
    for name, gx, wx, dur, vg, sg, lds in rows:
        agg[(name.split("(")[0][:110], gx, wx, vg, sg, lds)].append(dur)
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':112s} {'grid':>9s} {'wg':>4s} {'vgpr':>4s} {'sgpr':>4s} {'lds':>6s} {'calls':>5s} {'total_us':>12s} {'avg_us':>11s} {'min_us':>11s} {'max_us':>11s} {'%':>6s}")
    for (name, gx, wx, vg, sg, lds), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name:112s} {gx:9d} {wx:4d} {vg:4d} {sg:4d} {lds:6d} {len(v):5d} {sum(v)/1e3:12.1f} {sum(v)/len(v)/1e3:11.1f} {min(v)/1e3:11.1f} {max(v)/1e3:11.1f} {100*sum(v)/tot:6.2f}")


def pmc(db, sub=""):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection"))
    per = defaultdict(dict)
    meta = {}
    for did, name, grid, cname, val, dur in rows:
        if sub in name:
            per[did][cname] = per[did].get(cname, 0.0) + val
            meta[did] = (name.split("(")[0][:90], grid, dur)
    names = sorted({c for d in per.values() for c in d})
    print("dispatch grid duration_us " + " ".join(names) + "  kernel")
    for did in sorted(per):
        n, g, dur = meta[did]
        print(did, g, f"{dur/1e3:.1f}", " ".join(f"{per[did].get(c, float('nan')):.6g}" for c in names), n)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
