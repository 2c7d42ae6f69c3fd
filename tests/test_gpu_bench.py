"""bench.py on the GPU box: the N > 1 line is produced by N ranks bench.py starts itself, and the default line carries the
measurement (clock, what binds, measured traffic)."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=1500):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert pr.returncode == 0, pr.stderr[-3000:]
    return json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])


def test_two_ranks_started_by_bench_itself():
    """`python bench.py --gpus 2 --steps 1` WITHOUT torchrun (gloo: both ranks on the one GPU of this box): n_gpus is the number of
    ranks that computed the line, and the gathered Gram equals single-rank evaluations."""
    line = _bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--weak", "--no-cpu-baseline"], {"GPSIG_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == 2
    seen = line["ranks_seen"]
    assert seen["world_size"] == 2 and seen["backend"] == "gloo" and seen["launched_by"] == "bench.py itself"
    assert sorted(r["rank"] for r in seen["ranks"]) == [0, 1] and all(r["arch"].startswith("gfx950") for r in seen["ranks"])
    assert line["verify_max_abs_diff_vs_single_rank"] <= 1e-12
    assert line["rel_err"] <= 1e-6
    # where each rank's time went (round 5): the row-block kernels, the gathers, rank 0's symmetrisation
    pr = line["per_rank"]
    assert [t["rank"] for t in pr["ranks"]] == [0, 1] and all(t["compute_ms"] > 0 and t["chunks"] == 4 for t in pr["ranks"])
    assert pr["compute_max_over_min"] >= 1.0 and pr["symmetrise_ms_rank0"] > 0 and pr["ranks"][1]["symmetrise_ms"] == 0.0


def test_clock_probe_reads_a_plausible_shader_clock():
    import torch
    from gpsig_amd import _lib, kernels
    dev = torch.device("cuda:0")
    ctx = _lib.context(0, torch.cuda.current_stream(dev).cuda_stream)
    kern = kernels.SignatureLinear(64 * 8, 8, 5)
    X = torch.randn(1024, 64 * 8, dtype=torch.float64, device=dev)
    kern.K(X)
    torch.cuda.synchronize()
    ctx.clock_probe_start(200.0, 64)                        # far longer than the work: the read below ends it
    t0 = time.perf_counter()
    for _ in range(6):
        kern.K(X)
    torch.cuda.current_stream(dev).synchronize()
    busy_ms = (time.perf_counter() - t0) * 1e3
    mean, lo, hi, window = ctx.clock_probe_read()
    assert 1.0 < lo <= mean <= hi < 2.6, (mean, lo, hi)      # MI355X: 2.4 GHz peak engine clock, lower under float64 load
    assert 0.5 * busy_ms < window < busy_ms + 5.0, (window, busy_ms)
    with pytest.raises(ValueError):
        ctx.clock_probe_read()                               # one read per start
    ctx.clock_probe_start(2.0, 8)                            # a probe nobody reads leaves by itself
    torch.cuda.synchronize()


def test_default_line_carries_the_measurement():
    line = _bench(["--steps", "5", "--warmup", "2"])
    assert line["n_gpus"] == 1 and line["config"]["name"] == "c2" and line["rel_err"] <= 1e-6
    rf = line["roofline"]
    # the headline Gram is the feature contraction on the float64 matrix cores: priced in TFLOP/s against their peak
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 78.6
    assert 0.5 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["mfma"]["depth"] == 37456 and 0.7 < rf["mfma"]["kernel_share_of_step"] <= 1.0
    assert rf["issue_frac"] is None and rf["stream_frac"] > 0
    assert 1.0 < line["clock_ghz"] < 2.6
    assert rf["traffic"] and rf["traffic_source"]["how"].startswith("measured by this run")
    assert "sig_gram_dma_kernel" in rf["traffic_source"]["kernel"]
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    names = [s["name"] for s in line["secondary"]]
    assert names == ["c2-linear-lattice", "c2-linear-order5", "c2-rbf", "c3-rbf", "c3-rbf-increments", "c3-linear", "c5-rbf", "c4-single-gpu",
                     "grad-c2shape-n1024-linear", "grad-c2shape-n1024-linear-level-primitives", "grad-c2shape-n1024-linear-pair-kernels",
                     "grad-c2shape-n1024-rbf", "grad-c2shape-n1024-matern32", "grad-n512-l128-rbf", "c2-matern32", "grad-n512-rbf-order2", "grad-c3-rbf-order2",
                     "svgp-step-charactertrajectories", "svgp-step-netflow", "svgp-step-arabicdigits", "svgp-step-cmusubject16", "c3-svgp-predict",
                     "c2-rbf-order2", "c3-rbf-order2"]
    ho = line["secondary"][1]                           # the higher-order algorithm at order = num_levels: the same contraction, the same time
    assert ho["bound"] == "mfma" and ho["ms_per_step"] < 1.3 * line["ms_per_step"]
    lat = line["secondary"][0]                          # the same Gram through the pair recursion: vector-issue bound, about twice the time
    assert lat["bound"] == "valu-issue" and 0.3 < lat["issue_frac"] <= 1.05 and lat["ms_per_step"] > line["ms_per_step"]
    # the instruction counts behind issue_frac are measured by the run itself (one rocprofv3 --pmc SQ_INSTS_VALU pass per record), not typed in
    for i in (0, 2, 3, 4, 6):
        assert line["secondary"][i]["issue_source"].startswith("measured by this run"), line["secondary"][i]
    c4 = line["secondary"][7]                           # configs[3] on one GPU: the N = 1 point of the scaling series' own workload
    assert c4["name"] == "c4-single-gpu" and c4["rel_err"] <= 1e-6 and "N=32768" in c4["workload"] and c4["ms_per_step"] > 100
    for s in line["secondary"]:
        assert not s.get("error"), s
        assert s["ms_per_step"] > 0
        if not s["name"].startswith(("grad-", "svgp-step-", "c3-svgp")) and s["name"] != "c2-matern32":
            assert s["rel_err"] <= (1e-4 if s["dtype"] == "f32" else 1e-6) and s["clock_ghz"] > 1.0
    c3l = line["secondary"][5]                          # configs[2] with SignatureLinear: Kzx as one product of level features (round 4)
    assert c3l["bound"] == "mfma" and c3l["ms_per_step"] < line["secondary"][3]["ms_per_step"]
    c5 = line["secondary"][6]                           # configs[4]: priced as issue-bound from its counters (round 4)
    assert c5["bound"] == "valu-issue" and 0.5 < c5["issue_frac"] <= 1.1
    # round 6: one SVGP step at the reference's own run settings (benchmarks/run_gpsig_benchmarks.py:32) -- the wide shapes (10, 28, 126 columns) cost
    # no more per column-FMA than the 8-column shape the tile kernels serve -- and configs[2] end to end
    sv = {s["name"]: s for s in line["secondary"] if s["name"].startswith("svgp-step-")}
    base = sv["svgp-step-charactertrajectories"]["column_fma_per_s"]
    for name in ("svgp-step-netflow", "svgp-step-arabicdigits", "svgp-step-cmusubject16"):
        assert sv[name]["column_fma_per_s"] > base / 3.0, (name, sv[name]["column_fma_per_s"], base)
        assert sv[name]["ms_per_step"] < 40.0
    pr = [s_ for s_ in line["secondary"] if s_["name"] == "c3-svgp-predict"][0]
    assert pr["name"] == "c3-svgp-predict" and pr["finite"] and 0 < pr["covariances_ms"] < pr["ms_per_step"] < 30.0
    g = {s["name"]: s["ms_per_step"] for s in line["secondary"] if s["name"].startswith("grad-")}
    # the linear kernel's reverse pass through the feature contraction: several times faster than through the pair kernels
    assert g["grad-c2shape-n1024-linear"] * 3 < g["grad-c2shape-n1024-linear-pair-kernels"]
    assert g["grad-c2shape-n1024-linear"] < 1.1 * g["grad-c2shape-n1024-linear-level-primitives"]     # the level sum as one op is not slower
    # round 5's fused reverse kernel is what runs (the Lam-through-HBM route took 48 / 106 / 55 ms for these three)
    assert g["grad-c2shape-n1024-rbf"] < 25 and g["grad-c2shape-n1024-matern32"] < 35 and g["grad-n512-l128-rbf"] < 32
    assert [s_ for s_ in line["secondary"] if s_["name"] == "c2-matern32"][0]["ms_per_step"] < 75       # (run-time-kind instances: 77-84 ms)
    assert line["secondary"][3]["stream_frac"] > 0     # printed as stream_frac, never as an HBM fraction above 1
    # round 6: order 2 of SignatureRBF -- exact instances (K(X): 2.5 x order 1, was 9 x), higher-order chains in the tile kernel (Kzx: 1.15 x), fused reverse passes
    by = {s_["name"]: s_ for s_ in line["secondary"]}
    assert by["c2-rbf-order2"]["ms_per_step"] < 4.5 * by["c2-rbf"]["ms_per_step"] and by["c2-rbf-order2"]["rel_err"] <= 1e-6
    assert by["c3-rbf-order2"]["ms_per_step"] < 2.0 * by["c3-rbf"]["ms_per_step"] and by["c3-rbf-order2"]["rel_err"] <= 1e-6
    assert g["grad-n512-rbf-order2"] < 120 and g["grad-c3-rbf-order2"] < 6 * by["c3-rbf-order2"]["ms_per_step"] + 6
