#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a gfx950 assembly file (hipcc -S --cuda-device-only): for every backward branch the
number of float64 / other vector / scalar / LDS / global / scalar-load / scratch instructions between its target and itself.
    python tools/isa_loops.py file.s <mangled-or-demangled substring of the kernel name> [min_len]
This is what the per-step instruction budgets in profiles/ are read from."""
import re, subprocess, sys


def kernel_text(path, pat):
    t = open(path).read()
    for nm in re.findall(r"^(_Z[^\n:]*):", t, re.M):
        dm = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
        if pat in nm or pat in dm:
            i = t.index("\n" + nm + ":")
            return dm, t[i:t.index(".Lfunc_end", i)].split("\n")
    raise SystemExit("no kernel matching " + pat)


def classify(op):
    if op in ("v_readlane_b32", "v_writelane_b32"): return "lane_spill"
    if op.startswith("v_"):
        if "f64" in op or op.startswith("v_ldexp") : return "v_f64"
        if "dpp" in op: return "v_dpp"
        return "v_other"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "s_load"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    name, lines = kernel_text(path, pat)
    print("#", name)
    labels = {}
    for k, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = k
    for k, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.search(r"s_branch (\.LBB\d+_\d+)", l)
        if not (m and m.group(1) in labels and labels[m.group(1)] < k): continue
        a = labels[m.group(1)]
        cnt = {}
        dpp = 0
        for b in lines[a:k + 1]:
            b = b.strip()
            if not b or b[0] in ";.": continue
            c = classify(b.split()[0])
            if " row_" in b or " wave_" in b or "quad_perm" in b: c = "v_dpp" if c.startswith("v_") else c
            cnt[c] = cnt.get(c, 0) + 1
        n = sum(cnt.values())
        if n >= min_len:
            valu = sum(v for c, v in cnt.items() if c.startswith("v_"))
            print(f"lines {a}-{k}: {n} instructions, {valu} vector | " + "  ".join(f"{c} {v}" for c, v in sorted(cnt.items())))




def blocks(path, pat, min_len=60):
    """Basic blocks (between labels / branches) of at least min_len instructions: the straight-line step bodies of a kernel whose loop also
    holds conditionally executed blocks (pair boundaries)."""
    name, lines = kernel_text(path, pat)
    print("#", name)
    start = 0
    out = []
    for k, l in enumerate(lines):
        t = l.strip()
        is_label = bool(re.match(r"^(\.LBB\d+_\d+):", t)) or t.startswith("; %bb")
        is_branch = t.startswith("s_cbranch") or t.startswith("s_branch") or t.startswith("s_endpgm")
        if is_label and k > start:
            out.append((start, k - 1)); start = k
        if is_branch:
            out.append((start, k)); start = k + 1
    for a, b in out:
        cnt = {}
        for t in lines[a:b + 1]:
            t = t.strip()
            if not t or t[0] in ";.": continue
            c = classify(t.split()[0])
            if " row_" in t or " wave_" in t or "quad_perm" in t: c = "v_dpp" if c.startswith("v_") else c
            if t.startswith("v_mov_b64"): c = "v_mov64"
            if t.startswith("v_cndmask"): c = "v_cndmask"
            cnt[c] = cnt.get(c, 0) + 1
        n = sum(cnt.values())
        if n >= min_len:
            valu = sum(v for c, v in cnt.items() if c.startswith("v_"))
            print(f"lines {a}-{b}: {n} instructions, {valu} vector | " + "  ".join(f"{c} {v}" for c, v in sorted(cnt.items())))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--blocks":
    blocks(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 60)
    raise SystemExit(0)

if __name__ == "__main__":
    main()
