#!/usr/bin/env python3
"""Fixtures for the cases round 4's randomised sweeps reported above tolerance (profiles/r04_fuzz.txt), so that what the product does about
them is a committed test instead of a note in a log:

  * seed 51, case 187 of tools/fuzz_parity.py -- float64, SignatureLinear, order 3, ONE column, 33 / 32 observations, normalised: the float64
    ORACLE is 1.06e-4 (K(X, X2)) and 1.05e-6 (K(X)) away from an 80-bit evaluation of the same algorithm (numpy longdouble through the oracle's
    own code, `dtype=np.longdouble`): the index-tuple sums of the pair recursion cancel from ~1e8 to ~1e-3 for a one-column sequence at order 3.
    The product's feature route sums per sequence, does not cancel, and agrees with the 80-bit values -- so it departs from the float64
    restatement of the reference knowingly, in the accurate direction.  Stored: inputs, the oracle's float64 result, the 80-bit result.
  * the float32 cases of seed 51 with a one-column state space (num_features = 1; the class of every float32 miss the sweeps reported: 1.2e-4 ..
    5.6e-3 against 1e-4): stored with the float64 oracle's values.  Round 5 evaluates float32 requests on one-column state spaces in float64
    (gpsig_amd/kernels.py, _f32_upcast), so they are held to the float32 tolerance again.

Round 5's sweeps (seeds 71 and 72 with the rebuilt Kzx tile kernel, profiles/r05_fuzz.txt: 2,500 cases, 3 above tolerance, all float32) add a
second file, fuzz_cases_r5.npz:
  * seed 71, case 779 -- float32, SignatureLinear, order 5, ONE column, 16 / 90 observations: evaluated in float64 by the product (the class
    above) and 8.7e-7 from the 80-bit values, while the float64 ORACLE is 2.3e-3 away from them: the miss was the oracle's.  Stored with both.
  * seed 71 case 255 and seed 72 case 298 -- float32, SignatureCosine, inducing tensors against sequences of TWO / THREE observations,
    normalised: 1.7e-4 / 3.9e-4 on the matrix scale.  The cosine kernel's values are ratios of float32 inner products and the levels of such a
    short sequence are a handful of their double increments: float32 arithmetic, not a defect of a kernel.  Stored with the float64 oracle's
    values; the test states 1e-3 for this class.

Replays the sweep's random stream (tools/fuzz_parity.draw_case: no GPU needed); the 80-bit Grams take about two minutes each.
    python tests/golden/make_fuzz_cases.py           -> tests/golden/fuzz_cases.npz
    python tests/golden/make_fuzz_cases.py round5    -> tests/golden/fuzz_cases_r5.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as F  # noqa: E402

SEED, CASES = 51, 1500


def main():
    rng = np.random.default_rng(SEED)
    out = {}
    names = []
    for it in range(CASES):
        cs = F.draw_case(rng)
        want80 = it == 187
        # (small cases only, a dozen of them: the fixture file stays under a megabyte)
        f32_one_column = cs["f32"] and cs["d"] == 1 and cs["N1"] <= 40 and cs["T"] <= 70 and sum(n != "c187" for n in names) < 12
        if not (want80 or f32_one_column):
            continue
        key = "c%d" % it
        names.append(key)
        X, X2, Z = (cs[k].astype(np.float64) for k in ("Xq", "X2q", "Zq"))
        ko = F.oracle_for(cs)
        out[key + "_X"], out[key + "_X2"], out[key + "_Z"] = cs["Xq"], cs["X2q"], cs["Zq"]
        out[key + "_ls"], out[key + "_var"] = cs["kw"]["lengthscales"], cs["kw"]["variances"]
        out[key + "_meta"] = np.array([cs["M"], cs["order"], cs["d"], cs["lags"], cs["L1"], cs["L2"], int(cs["norm"]), int(cs["diff"]), int(cs["f32"]),
                                       int(cs["incr"]), cs["T"]], dtype=np.int64)
        out[key + "_base"] = np.array(cs["base"])
        out[key + "_K"] = ko.K(X)
        out[key + "_Kx"] = ko.K(X, X2)
        if want80:
            k80 = F.oracle_for(cs, np.longdouble)
            out[key + "_K80"] = np.asarray(k80.K(X.astype(np.longdouble)), dtype=np.float64)
            out[key + "_Kx80"] = np.asarray(k80.K(X.astype(np.longdouble), X2.astype(np.longdouble)), dtype=np.float64)
        else:
            out[key + "_Kdiag"] = ko.Kdiag(X)
            out[key + "_Kzx"] = ko.K_tens_vs_seq(Z, X, increments=cs["incr"])
            out[key + "_Kzz"] = ko.K_tens(Z, increments=cs["incr"])
        print(key, cs["desc"], flush=True)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fuzz_cases.npz"), **out)
    print("wrote", len(names), "cases")


def round5():
    out, names = {}, []
    for seed, picks in ((71, {255: "kzx", 779: "kx80"}), (72, {298: "kzx"}), (73, {629: "kx", 736: "k80"})):
        rng = np.random.default_rng(seed)
        for it in range(max(picks) + 1):
            cs = F.draw_case(rng)
            if it not in picks:
                continue
            key = "s%dc%d" % (seed, it)
            names.append(key)
            X, X2, Z = (cs[k].astype(np.float64) for k in ("Xq", "X2q", "Zq"))
            ko = F.oracle_for(cs)
            out[key + "_X"], out[key + "_X2"], out[key + "_Z"] = cs["Xq"], cs["X2q"], cs["Zq"]
            out[key + "_ls"], out[key + "_var"] = cs["kw"]["lengthscales"], cs["kw"]["variances"]
            out[key + "_meta"] = np.array([cs["M"], cs["order"], cs["d"], cs["lags"], cs["L1"], cs["L2"], int(cs["norm"]), int(cs["diff"]), int(cs["f32"]),
                                           int(cs["incr"]), cs["T"]], dtype=np.int64)
            out[key + "_base"] = np.array(cs["base"])
            if picks[it] == "kzx":
                out[key + "_Kzx"] = ko.K_tens_vs_seq(Z, X, increments=cs["incr"])
            elif picks[it] == "k80":
                out[key + "_K"] = ko.K(X)
                out[key + "_K80"] = np.asarray(F.oracle_for(cs, np.longdouble).K(X.astype(np.longdouble)), dtype=np.float64)
            elif picks[it] == "kx":
                out[key + "_Kx"] = ko.K(X, X2) if cs["L1"] == cs["L2"] else F._cross(ko, X, X2, cs["d"])
            else:
                cross = (lambda k, a, b: k.K(a, b)) if cs["L1"] == cs["L2"] else (lambda k, a, b: F._cross(k, a, b, cs["d"]))
                out[key + "_Kx"] = cross(ko, X, X2)
                out[key + "_Kx80"] = np.asarray(cross(F.oracle_for(cs, np.longdouble), X.astype(np.longdouble), X2.astype(np.longdouble)), dtype=np.float64)
            print(key, cs["desc"], flush=True)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fuzz_cases_r5.npz"), **out)
    print("wrote", len(names), "cases")


if __name__ == "__main__":
    round5() if sys.argv[1:2] == ["round5"] else main()
