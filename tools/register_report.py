#!/usr/bin/env python3
"""Compile every translation unit of gpsig_amd/csrc to gfx950 assembly and list each kernel's registers, scratch and occupancy as the
compiler reports them (; NumVgprs / ; ScratchSize / ; Occupancy).  The committed table (profiles/rNN_register_report.txt) is what a later
change to a shared header is diffed against: one register more in a kernel that sits on an occupancy step costs a wavefront per SIMD or a
workgroup per CU without failing any test (round 4: low-rank mode 2.8 -> 4.6 ms from a helper added to base_eval).
    python tools/register_report.py [-j 8] [pattern] > profiles/rNN_register_report.txt
    python tools/register_report.py --diff old.txt new.txt"""
import glob, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpsig_amd", "csrc")


def report(tu, tmp):
    out = os.path.join(tmp, os.path.basename(tu) + ".s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, tu],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = []
    if r.returncode != 0 or not os.path.exists(out):
        return [(os.path.basename(tu), "COMPILE FAILED", 0, 0, 0)]
    text = open(out).read()
    for m in re.finditer(r"^(_Z[^\n:]*):", text, re.M):
        q = re.search(r"; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", text[m.start():m.start() + 400000], re.S)
        nxt = re.search(r"^_Z[^\n:]*:", text[m.end():], re.M)
        if q and (nxt is None or q.start() < nxt.start() + (m.end() - m.start())):
            try:
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip() or m.group(1)
            except Exception:
                name = m.group(1)
            rows.append((os.path.basename(tu), name.split("(")[0][:120], int(q.group(1)), int(q.group(2)), int(q.group(3))))
    os.remove(out)
    return rows


def load(fn):
    out = {}
    for l in open(fn):
        f = l.rstrip("\n").split("\t")
        if len(f) == 5 and f[2].isdigit():
            out[(f[0], f[1])] = tuple(int(x) for x in f[2:])
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--diff":
        a, b = load(sys.argv[2]), load(sys.argv[3])
        for k in sorted(set(a) | set(b)):
            if a.get(k) != b.get(k) and (a.get(k) is None or b.get(k) is None or a[k][1:] != b[k][1:]):
                print(k[0], k[1], a.get(k), "->", b.get(k))
        return
    jobs, pat = 8, ""
    args = sys.argv[1:]
    if args and args[0] == "-j":
        jobs, args = int(args[1]), args[2:]
    if args:
        pat = args[0]
    tus = sorted(t for t in glob.glob(os.path.join(SRC, "*.hip")) if pat in os.path.basename(t))
    print("# translation unit\tkernel\tNumVgprs\tScratchSize\tOccupancy   (hipcc --offload-arch=gfx950 -O3 -S; tools/register_report.py)")
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(jobs) as ex:
        for rows in ex.map(lambda t: report(t, tmp), tus):
            for r in rows:
                print("\t".join(str(x) for x in r), flush=True)


if __name__ == "__main__":
    main()
