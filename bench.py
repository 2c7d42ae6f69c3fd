#!/usr/bin/env python3
"""Benchmark of the signature-kernel evaluation path on MI355X: sequence-pairs/s of SignatureKernel.K (and Kzx).

    python bench.py [--gpus N --steps K --warmup W] [--config c2|c3|c4|c5] [--base linear|rbf]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (BASELINE.json configs; the names C1..C5 are SURVEY.md section 8's):
  c2  configs[1]: full N x N Gram, N=4096, L=64, d=8, num_levels=5, fp64, SignatureLinear (--base rbf: SignatureRBF), 1 GPU.
      The default at --gpus 1: the configuration BASELINE.json's metric is quoted on.
  c4  configs[3]: the same Gram with N=32768.  The default at --gpus N > 1: every rank computes the entries its rows own
      (gpsig_kernel_K_symm_rows_compact) in chunks, each chunk's asynchronous RCCL gather to rank 0 overlaps the next chunk's
      computation, rank 0 symmetrises (gpsig_amd/parallel.py).  The problem is the same for N = 2, 4, 8 ("strong" scaling
      among them; the N = 1 line of the driver's series is c2, and sequence-pairs/s is comparable across both because the
      pair kernel's rate does not depend on N).  --weak restores per-GPU-constant work: N_total = 4096 * sqrt(N).
      With --gpus 1 the whole N=32768 Gram is evaluated on one GPU.
  c3  configs[2]: SVGP inducing-tensor path, Kzz + Kzx + Kxx-diag (K_tens_n_seq_covs), T=512 inducing tensors, N=16384, L=50,
      d=6, num_levels=4, fp64, SignatureRBF (--increments: Z holds increments).  A pair is one (tensor, sequence) entry.
      With --gpus N > 1 the sequences are split over the ranks (parallel.ShardedCovs; 3 ms of work: a functional path, not a
      scaling benchmark).
  c5  configs[4]: N=2048, L=128, d=16, num_levels=6, fp32, SignatureRBF, full Gram.
One step = one complete evaluation with the inputs already resident in HBM and the result left in HBM.  Prints ONE JSON line
on rank 0: the driver's contract fields + `roofline` (what binds the dominant kernel -- `bound`, `issue_frac`, `alu_frac` --, the
pair-stream fraction of SURVEY 8(d) as `stream_frac`, the kernel's HIP-event time on the library's stream, the HBM traffic
measured by two rocprofv3 --pmc passes of this very command), `clock_ghz` (the shader clock during the timed region, sampled
on the device), `cpu_baseline` (the oracle's op-for-op restatement of the reference's TF graph on the host cores, bounded
sample), `rel_err` (a sub-sample of the timed output against the oracle, outside the timed region),
`end_to_end_ms_host_pointers` (the same evaluation from and to host memory: H2D + compute + D2H) and, on the default
single-GPU line, `secondary`: the other single-GPU configurations (c2 through the pair recursion, c2 at order 5, c2 RBF, c3, c3 with increments, c3 with SignatureLinear, c5), 3 warm-up + 10 steps each.

--gpus N > 1 without a torchrun environment launches the N ranks itself (python -m torch.distributed.run, one process per
GPU, rendezvous on 127.0.0.1) and fails when the node has fewer than N GPUs: a line with "n_gpus": N was computed by N ranks,
listed in `ranks_seen`.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6        # vector fp64 (half the fp32 vector rate)
FP32_VECTOR_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: vector fp32
METRIC = "sequence-pairs/sec for SignatureKernel.K (N,L,d,sig_level); fp64 rel-err vs ref"

#                N      L    d   M   T    dtype   base      data
WORKLOADS = {
    "c2": dict(N=4096, L=64, d=8, M=5, T=0, dtype="f64", base="linear", data="white"),
    "c4": dict(N=32768, L=64, d=8, M=5, T=0, dtype="f64", base="linear", data="white"),
    "c3": dict(N=16384, L=50, d=6, M=4, T=512, dtype="f64", base="rbf", data="walk"),
    "c5": dict(N=2048, L=128, d=16, M=6, T=0, dtype="f32", base="rbf", data="walk"),
}
BASELINE_INDEX = {"c2": 1, "c3": 2, "c4": 3, "c5": 4}


def make_inputs(w, n=None):
    """Seeded synthetic inputs: white noise as the notebook's cell 4, or random walks of step 0.2 / 0.1."""
    n = n or w["N"]
    rng = np.random.default_rng(0)
    if w["data"] == "white":
        X = rng.standard_normal((n, w["L"], w["d"]))
    else:
        X = np.cumsum((0.2 if w["T"] else 0.1) * rng.standard_normal((n, w["L"], w["d"])), axis=1)
    return X.reshape(n, -1)


def make_tensors(w, increments):
    rng = np.random.default_rng(1)
    lt = w["M"] * (w["M"] + 1) // 2
    return rng.standard_normal((lt, w["T"], 2, w["d"]) if increments else (lt, w["T"], w["d"]))


def lengthscales(w):
    return 1.0 if w["base"] == "linear" else math.sqrt(w["d"])


def stream_bytes_per_pair(w, increments=False):
    """ALGORITHMIC bytes per pair, pair-stream model of SURVEY.md 8(d): a pair loads its two streams and writes its result."""
    s = 8 if w["dtype"] == "f64" else 4
    if w["T"]:
        lt = w["M"] * (w["M"] + 1) // 2
        return w["L"] * w["d"] * s + (2 if increments else 1) * lt * w["d"] * s + s
    return 2 * w["L"] * w["d"] * s + s


def flops_per_pair(w, increments=False):
    """(reference op count, flops the kernels execute) per pair, SURVEY.md 8(d)."""
    L, d, M = w["L"], w["d"], w["M"]
    if w["T"]:
        lt = M * (M + 1) // 2
        npts = lt * (2 if increments else 1)
        ref = npts * (2 * L * d + 3 * L) + lt * (L - 1) + 2 * (L - 1) * (M * (M - 1) // 2) + M * (L - 1)
        return ref, ref
    ref = 2 * L * L * d + (L - 1) * (L - 1) * 4 * M + (4 * L * L if w["base"] == "rbf" else 0)
    if w["base"] == "linear":
        ex = (L - 1) * (L - 1) * (2 * d + 3 * M - 1)                    # increments first, then the row sweep
    else:
        ex = L * L * (2 * d + 4) + (L - 1) * (L - 1) * (3 + 3 * M - 1)  # kappa on points (+ one exp each), double increment, row sweep
    return ref, ex


# ---- CPU baseline: the oracle (test infrastructure) timed on the host cores -- reported, never shipped or measured as product
def cpu_tile(args):
    import numpy as _np
    from oracle import sigkern_oracle as O
    cfg, tile, increments = args
    w = WORKLOADS[cfg]
    kern = O.SignatureKernelOracle(w["L"] * w["d"], w["d"], w["M"], base=w["base"], normalization=False, lengthscales=None)
    X = make_inputs(w, tile).reshape(tile, w["L"], w["d"])
    if w["T"]:
        rng = _np.random.default_rng(1)
        lt = w["M"] * (w["M"] + 1) // 2
        Z = rng.standard_normal((lt, tile, 2, w["d"]) if increments else (lt, tile, w["d"]))
        t0 = time.perf_counter()
        kern._K_tens_vs_seq(Z, X, increments=increments)
    else:
        t0 = time.perf_counter()
        kern._K_seq(X, X)
    return time.perf_counter() - t0


def host_cpus():
    """CPUs this process can actually use: the cgroup quota where there is one (the GPU boxes run the container with
    cpu.max = 16 CPUs on a 256-thread host; oversubscribing it with one thread per visible CPU throttles everything)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except Exception:
            continue
    return n


def cpu_baseline_numpy(cfg, increments, budget_s=8.0):
    """The reference's TF-CPU graph restated op for op in NumPy (oracle/sigkern_oracle.py), on tiles of sequences at the benchmark
    shape, spread over single-threaded worker processes.  Bounded sample."""
    import multiprocessing as mp
    w = WORKLOADS[cfg]
    workers = max(1, min(64, host_cpus()))
    tile = 64 if w["T"] else (32 if w["L"] <= 64 else 16)
    ctx = mp.get_context("spawn")
    for v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(v, "1")
    with ctx.Pool(workers) as pool:
        t1 = pool.map(cpu_tile, [(cfg, tile, increments)] * workers)          # warm-up + calibration
        rounds = max(1, int(budget_s / max(float(np.median(t1)), 1e-3)))
        ntiles = workers * rounds
        t0 = time.perf_counter()
        pool.map(cpu_tile, [(cfg, tile, increments)] * ntiles)
        wall = time.perf_counter() - t0
    return {"value": ntiles * tile * tile / wall, "unit": "sequence-pairs/s", "cores": workers,
            "sample": f"{ntiles} tiles of {tile}x{tile} pairs, whole-tensor NumPy ops (matmul, 4-slice difference, 2 cumsums + multiply + "
                      f"reduce per level), {workers} single-threaded worker processes, {wall:.1f} s wall"}


def cpu_baseline(cfg, base, increments, budget_s=10.0, numpy_leg=True):
    """cpu_baseline of the bench line: the oracle's C restatement of the reference's graph (oracle/sigkern_ref.c: kappa lattice,
    double difference, per level two exclusive cumsums + multiply + reduce -- gpsig/kernels.py:225-230, signature_algs.py:25-35 /
    :114-125 -- one pair at a time so that a lattice stays in cache, OpenMP over pairs on every host thread), on a bounded sample
    of the benchmark shape; `numpy` beside it is the whole-tensor NumPy restatement the reference's TensorFlow ops map to one to
    one.  kind "port": TensorFlow 1.15 is not installable here."""
    os.environ.setdefault("OMP_NUM_THREADS", str(host_cpus()))      # before libgomp starts: one thread per usable CPU
    from oracle import cref
    w = WORKLOADS[cfg]
    rng = np.random.default_rng(0)
    L, d, M = w["L"], w["d"], w["M"]
    threads = cref.threads()
    if w["T"]:
        lt = M * (M + 1) // 2
        t_t, n_t = 64, 64 * max(1, threads // 8)
        X = np.cumsum(0.2 * rng.standard_normal((n_t, L, d)), axis=1)
        Z = rng.standard_normal((lt, t_t, 2, d) if increments else (lt, t_t, d))
        call = lambda: cref.tens_vs_seq_levels(Z, X, M, base)                      # noqa: E731
        pairs_call = t_t * n_t
        what = f"Kzx levels of {t_t} inducing tensors x {n_t} sequences"
    else:
        n_t = 64 * max(1, int(round(math.sqrt(threads))))
        X = rng.standard_normal((n_t, L, d)) if w["data"] == "white" else np.cumsum(0.1 * rng.standard_normal((n_t, L, d)), axis=1)
        call = lambda: cref.seq_levels(X, X, M, base)                              # noqa: E731
        pairs_call = n_t * n_t
        what = f"all {n_t}x{n_t} sequence pairs of {n_t} sequences (levels 0..{M})"
    call()                                                                          # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    call()
    t1 = time.perf_counter() - t0
    reps = max(1, int(budget_s / max(t1, 1e-3)))
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    wall = time.perf_counter() - t0
    res = {"value": reps * pairs_call / wall, "unit": "sequence-pairs/s", "cores": threads, "kind": "port",
           "sample": f"{reps} x {what} at L={L}, d={d}, num_levels={M}, {base}, fp64, oracle/sigkern_ref.c (gcc -O3 -fopenmp, "
                     f"{threads} OpenMP threads), {wall:.1f} s wall"}
    res["implementation"] = "C restatement, OpenMP"
    if not numpy_leg:
        return res
    try:
        alt = cpu_baseline_numpy(cfg, increments)
        alt["implementation"] = "NumPy whole-tensor ops, one process per core"
    except Exception as e:                                                          # the NumPy leg is a side note: never lose the line over it
        res["other"] = {"error": repr(e)}
        return res
    # `value` is the faster of the two restatements on this host (the C one for the pair lattices; vectorised exp makes the
    # NumPy one the faster for the RBF tensor-vs-sequence chains); the other is kept beside it
    if alt["value"] > res["value"]:
        alt["kind"] = "port"
        res, alt = alt, res
        alt.pop("kind", None)
    res["other"] = alt
    return res


def rel_err(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float((np.abs(got - want) / (np.abs(want) + 1e-6 * np.abs(want).max())).max())


def oracle_rel_err(cfg, w, base, increments, Xh, Zh, out):
    """max |K - K_ref| / (|K_ref| + 1e-6 max|K_ref|) on a sub-sample of the timed output against the oracle."""
    from oracle import sigkern_oracle as O
    ko = O.SignatureKernelOracle(w["L"] * w["d"], w["d"], w["M"], base=base, lengthscales=lengthscales(dict(w, base=base)), order=w.get("order", 1))
    n = Xh.shape[0]
    if w["dtype"] == "f32":          # the oracle sees the inputs the kernel saw
        Xh = Xh.astype(np.float32).astype(np.float64)
        Zh = Zh.astype(np.float32).astype(np.float64) if Zh is not None else None
    if w["T"]:
        ns, ts = 24, 16
        want = ko.K_tens_n_seq_covs(Zh[:, :ts], Xh[:ns], increments=increments)
        Kzz, Kzx, Kxx = out
        return max(rel_err(Kzz[:ts, :ts].cpu().numpy(), want[0]), rel_err(Kzx[:ts, :ns].cpu().numpy(), want[1]),
                   rel_err(Kxx[:ns].cpu().numpy(), want[2]))
    k = 8 if w["L"] > 64 else 12
    idx = np.concatenate([np.arange(0, k), np.arange(n // 2 - k // 2, n // 2 + k // 2), np.arange(n - k, n)])   # diagonal, tie, wrap-around
    want = ko.K(Xh[idx])
    import torch
    ti = torch.as_tensor(idx, device=out.device)
    return rel_err(out[ti][:, ti].cpu().numpy(), want)


# ---- launching the ranks of an N > 1 run ------------------------------------------------------------------------------
def rank_launch_command(n, argv):
    """The command that runs this script on n ranks of one node: one process per GPU under torch.distributed.run; --standalone
    lets the c10d rendezvous pick its own port (no bind-and-release race), on 127.0.0.1 (the container's host name may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
            f"--nproc-per-node={int(n)}", os.path.abspath(__file__)] + list(argv)


def launch_ranks(n, argv):
    """python bench.py --gpus N without a torchrun environment: start the ranks and hand their exit code on."""
    import subprocess
    return subprocess.call(rank_launch_command(n, argv), env=dict(os.environ, GPSIG_BENCH_LAUNCHED="1"))


def visible_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def rank_identity(rank, local_rank, dev_index, backend):
    """What a rank reports into `ranks_seen`: its device as HIP sees it."""
    import torch
    ent = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(), "backend": backend, "device": None}
    if torch.cuda.is_available() and dev_index is not None:
        pr = torch.cuda.get_device_properties(dev_index)
        ent.update({"device": int(dev_index), "name": pr.name, "arch": getattr(pr, "gcnArchName", None),
                    "pci_bus_id": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
                    "uuid": str(getattr(pr, "uuid", ""))})
    return ent


# ---- counters of the dominant kernel: rocprofv3 --pmc passes of this command (MI355X_MICROARCH.md, "HBM" / "rocprofv3 PMC slots") ------------
def measure_pmc(cfg, base, increments, counters, lattice=False, timeout_s=240):
    """One rocprofv3 --pmc pass per counter (FETCH_SIZE and WRITE_SIZE cannot share a pass -- TCC slots -- and a pass never rides with a
    trace) of `bench.py --config ... --steps 2 --warmup 1 --timed-loop-only`.  Returns {counter: median over the dominant kernel's
    dispatches (summed over the counter's instances), "kernel", "grid", "dispatches"}, or None where rocprofv3 is missing or already
    wraps this process."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe or any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ):
        return None
    got = {}
    meta = None
    for counter in counters:
        tmp = tempfile.mkdtemp(prefix="gpsig_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "-d", tmp, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--config", cfg, "--base", base,
               "--steps", "2", "--warmup", "1", "--timed-loop-only"] + (["--increments"] if increments else []) + (["--lattice"] if lattice else [])
        try:
            pr = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True,
                                  env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
            try:
                pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, 9)
                return None
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(tmp) for f in fs if f.endswith(".db")]
            if not dbs:
                return None
            rows = list(sqlite3.connect(dbs[0]).cursor().execute(
                "select dispatch_id, kernel_name, grid_size, value, duration from counters_collection where counter_name = ?", (counter,)))
            per, info = {}, {}
            for did, name, grid, val, dur in rows:
                if "gpsig" not in name:
                    continue
                per[did] = per.get(did, 0.0) + val
                info[did] = (name.split("(")[0], int(grid), dur)
            tot = {}
            for did, (name, grid, dur) in info.items():
                tot[(name, grid)] = tot.get((name, grid), 0.0) + dur
            if not tot:
                return None
            kname, kgrid = max(tot, key=tot.get)                      # the dominant kernel: the largest share of device time
            vals = sorted(v for did, v in per.items() if info[did][:2] == (kname, kgrid))
            got[counter] = vals[len(vals) // 2]
            durs = sorted(info[did][2] for did in per if info[did][:2] == (kname, kgrid))
            got.setdefault("durations_ns", []).extend(durs)
            meta = (kname, kgrid, len(vals))
        except Exception:
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    durs = sorted(got.pop("durations_ns", []))
    got.update({"kernel": meta[0], "grid": meta[1], "dispatches": meta[2],
                # the dominant kernel's duration as the PROFILER saw it (median / mean over these passes' dispatches): a profiled launch runs at
                # a lower clock than an un-profiled one (MI355X_MICROARCH.md, DVFS), so fractions from it are the pessimistic end
                "kernel_us_profiled_median": (durs[len(durs) // 2] * 1e-3) if durs else None,
                "kernel_us_profiled_mean": (sum(durs) / len(durs) * 1e-3) if durs else None})
    return got


def measure_traffic(cfg, base, increments, lattice=False):
    """HBM traffic of the dominant kernel.  FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of 16-byte-per-lane loads at 64
    bytes; every read of these kernels is such a load or an LDS-DMA of that width), WRITE_SIZE is taken as is (KiB both)."""
    got = measure_pmc(cfg, base, increments, ("FETCH_SIZE", "WRITE_SIZE"), lattice)
    if not got:
        return None
    return {"fetch_size_kib": got["FETCH_SIZE"], "write_size_kib": got["WRITE_SIZE"],
            "bytes_per_launch": got["FETCH_SIZE"] * 2.0 * 1024.0 + got["WRITE_SIZE"] * 1024.0,
            "kernel": got["kernel"], "grid": got["grid"], "dispatches": got["dispatches"],
            "kernel_us_profiled_median": got["kernel_us_profiled_median"], "kernel_us_profiled_mean": got["kernel_us_profiled_mean"]}


# issue_frac = how much of the SIMDs' issue time the dominant kernel's vector instructions fill at the clock measured during the run: float64
# and DPP instructions take a SIMD's issue port for 4 cycles per wave64 instruction.  The instruction count is MEASURED BY THE RUN (round 5): one
# rocprofv3 --pmc SQ_INSTS_VALU pass of the same command (measure_pmc), so a kernel change cannot leave a stale constant behind (rounds 2-4
# priced the kernels with counts typed in from one profile pass).  No pass (rocprofv3 missing, already profiled, N > 1): issue_frac is null.
PAIRS_PER_WAVE = {"c5": 2}               # seq_pk2_kernel at G = 64: one pair group per wavefront, two y sequences packed (default 4: G = 16)
SIMDS = 1024


def depth_pieces(n, nslab):
    """Workgroups per tile along the contraction's depth for a symmetric Gram of n sequences: the library's model (csrc/api.hip,
    sig_features_K), restated here only to price the partial sums in `algorithmic_bytes_per_launch`."""
    nt = (n + 127) // 128
    tiles = nt * (nt + 1) // 2
    rb = 8.0 * n * n * 0.5
    best, pieces = 1e300, 1
    for ns in range(1, 129):
        if ns * 8 > nslab + 7:
            break
        w, pp = tiles * ns, nslab / ns
        t = (pp * 1.8e-6 + 10e-6 if w <= 256 else math.ceil(w / 512) * (pp * 3.6e-6 + 10e-6)) + (2 * ns * rb / 3e12 if ns > 1 else 0)
        if t < best * 0.999:
            best, pieces = t, ns
    return max(pieces, 4) if (tiles > 1024 and nslab >= 32) else pieces


FP64_MATRIX_PEAK_TFLOPS = 78.6          # MI355X_MICROARCH.md: dense float64 MFMA peak (= the vector peak) at 2.4 GHz


def run_workload(cfg, base, increments, steps, warmup, dev, rank=0, world=1, chunks=4, weak=False, checks=True, host_e2e=True,
                 traffic="measure", lattice=False, order=1, issue_src="measure"):
    """Times `steps` evaluations of one BASELINE configuration on this rank's device (inputs resident in HBM) and returns the
    bench-line fields on rank 0 (None elsewhere)."""
    import torch
    import torch.distributed as dist
    from gpsig_amd import _lib, kernels, parallel

    w = dict(WORKLOADS[cfg])
    w["base"] = base
    w["order"] = int(order)         # 1: the first-order algorithm (signature_algs.py:8-35); > 1: the higher-order one (:37-74)
    n_gpus = world
    if n_gpus > 1 and (weak or cfg == "c2"):
        w["N"] = int(round(4096 * math.sqrt(n_gpus) / 64.0)) * 64
    N, L, D, M, T = w["N"], w["L"], w["d"], w["M"], w["T"]
    tdt = torch.float64 if w["dtype"] == "f64" else torch.float32
    Xh = make_inputs(w)                                              # same data on every rank
    X = torch.as_tensor(Xh, device=dev).to(tdt)
    Zh = make_tensors(w, increments) if T else None
    Z = torch.as_tensor(Zh, device=dev).to(tdt) if T else None
    cls = kernels.SignatureLinear if base == "linear" else kernels.SignatureRBF
    kern = cls(L * D, D, M, lengthscales=lengthscales(w), order=int(order))
    gram = parallel.ShardedGram(kern, N, dev, rank, world, chunks=chunks) if not T else None
    covs = parallel.ShardedCovs(kern, N, dev, rank, world) if T else None      # world == 1: kern.K_tens_n_seq_covs itself

    def step():
        if T:
            return covs(Z, X, increments=increments)
        return gram(X)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ctx = _lib.context(dev.index or 0, torch.cuda.current_stream(dev).cuda_stream)
    # lattice=True: SignatureLinear through the pair recursion (round 1-2's kernel) instead of the feature contraction
    ctx.set_option("sig_features", 0 if lattice else -1)
    t_w = time.perf_counter()
    for _ in range(warmup):
        step()
    barrier()
    est_ms = (time.perf_counter() - t_w) * 1e3 / max(warmup, 1) * steps if warmup else 0.0
    if not warmup:                                                   # no estimate of the timed region: one untimed step gives it
        t_w = time.perf_counter()
        step()
        barrier()
        est_ms = (time.perf_counter() - t_w) * 1e3 * steps
    ctx.timing_reset()
    barrier()
    probe = False
    try:        # the shader clock over the timed region: a sleeping wavefront on a stream of its own, told to leave below
        ctx.clock_probe_start(max(1.0, min(est_ms * 1.2, 30000.0)), 128)
        probe = True
    except Exception:
        probe = False
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    clock = None
    if probe:   # before the device-wide synchronisation of barrier(), which would wait for the probe's own deadline
        try:
            torch.cuda.current_stream(dev).synchronize()
            g_mean, g_min, g_max, cov_ms = ctx.clock_probe_read()
            clock = {"mean": g_mean, "min": g_min, "max": g_max, "window_ms": cov_ms,
                     "per_xcd": [{"xcd": x, "ghz": g} for x, g in ctx.clock_probe_xcds()],
                     "how": "one sleeping wavefront per XCD on a stream of its own (mean = average over them): s_memtime (shader cycles) against s_memrealtime (100 MHz), readings "
                            "spread over the timed region (tools/clockcheck.hip checks the counter against an issue-bound loop)"}
        except Exception:
            clock = None
    barrier()
    dt = time.perf_counter() - t0
    if clock:
        clock["timed_region_ms"] = dt * 1e3
    kernel_ms, launches, _ = ctx.timing_get()      # HIP events around the dominant kernel, on the stream it was launched on
    timed_kernel, mfma_flops = ctx.timing_info()   # "sig_gram_dma_kernel" + its matrix-core flops when the feature contraction ran
    ctx.set_option("sig_features", -1)

    rank_times = None
    if world > 1:
        tt = torch.tensor([dt, kernel_ms / max(launches, 1)], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, per_launch_ms = float(tt[0].item()), float(tt[1].item())
        # where the LAST step's time went on every rank (ShardedGram.timings: stream time of its chunks' kernels, of the waits for the
        # gathers, of rank 0's symmetrisation) -- so that a scaling curve says whether balance or the gather bends it
        mine = gram.timings() if gram is not None else None
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        rank_times = gathered
    else:
        per_launch_ms = kernel_ms / max(launches, 1)
    launches_per_step = launches / max(steps, 1)
    if rank != 0:
        if checks and n_gpus > 1:
            pass
        return None

    pairs = float(T) * N if T else float(N) * N                    # entries delivered per step
    value = pairs * steps / dt
    b_pair = stream_bytes_per_pair(w, increments)
    f_ref, f_exec = flops_per_pair(w, increments)
    # one launch of the dominant kernel: this rank's share of a step, divided over the launches it took
    pairs_launch = pairs / n_gpus / max(launches_per_step, 1)
    evaluated_launch = pairs_launch if T else pairs_launch * (N + 1) / (2.0 * N)     # symmetric Gram: each unordered pair once
    achieved = pairs_launch * b_pair / (per_launch_ms * 1e-3) / 1e9
    alu_peak = FP64_VECTOR_PEAK_TFLOPS if w["dtype"] == "f64" else FP32_VECTOR_PEAK_TFLOPS
    tflops_exec = evaluated_launch * f_exec / (per_launch_ms * 1e-3) / 1e12
    stream_frac = achieved / HBM_PEAK_GBS
    alu_frac = tflops_exec / alu_peak
    ghz = clock["mean"] if clock else None
    issue = None
    if issue_src == "measure" and timed_kernel is None and n_gpus == 1 and cfg != "c4" and ghz and launches:
        m = measure_pmc(cfg, base, increments, ("SQ_INSTS_VALU",), lattice)
        if m:
            vpl = m["SQ_INSTS_VALU"]
            issue = {"valu_per_launch": vpl, "cycles_per_valu": 4, "simds": SIMDS,
                     "issue_frac": vpl * 4.0 / (SIMDS * ghz * 1e9 * per_launch_ms * 1e-3),
                     "source": "measured by this run: rocprofv3 --pmc SQ_INSTS_VALU, median over the dispatches of %s (grid %d)" % (m["kernel"], m["grid"])}
            if not T:
                pairs_per_wave = PAIRS_PER_WAVE.get(cfg, 4)              # G = 16: four pair groups per wavefront
                wave_steps = evaluated_launch / pairs_per_wave * L       # L lattice rows (L - 1 increments + the boundary row) per pair
                issue["wave_steps_per_launch"] = wave_steps
                issue["valu_per_wave_step"] = vpl / wave_steps           # (pair-boundary blocks, prologue and epilogue amortised over the steps)
    if T:
        bound = "valu-issue"
        binding = ("vector issue: the table-driven exps of the base kernel (10 per step and wave with increments) and the chain FMAs; the sequence "
                   "records are staged once per 64 tensors, so HBM sees ~2 % of the pair-stream bytes (DESIGN.md section 4)")
    elif w["dtype"] == "f64":
        bound = "valu-issue"
        binding = "float64 vector-ALU issue (instructions x 4 cycles per wave; DESIGN.md section 4)"
    elif issue:
        # (round 4: the counters of profiles/r04_pmc_c5.txt show the vector ALU active 0.9-0.97 of the launch -- SQ_ACTIVE_INST_VALU x 4 /
        # 1024 SIMDs against the kernel's cycles -- so this kernel, too, is bound by its instruction count, not by latency)
        bound = "valu-issue"
        binding = ("float32 vector-ALU issue: 118 vector instructions per wave-step (58 of them packed v_pk_fma_f32 at 4 cycles, the rest "
                   "hand-overs, selects and address arithmetic), DESIGN.md section 4")
    else:
        bound = "valu-latency"
        binding = "float32 dependent-instruction latency at the kernel's wavefronts per SIMD (DESIGN.md section 2.4)"
    mfma = None
    if timed_kernel in ("sig_gram_kernel", "sig_gram_dma_kernel"):
        fl_launch = mfma_flops / max(launches, 1)
        F = sum(D ** m for m in range(1, M + 1))
        ld = (F + 1 + 15) // 16 * 16
        tf = fl_launch / (per_launch_ms * 1e-3) / 1e12
        bound = "mfma"
        binding = ("float64 matrix cores: the Gram is ONE contraction of depth sum_m d^m + 1 = %d (explicit signature-level features, "
                   "v_mfma_f64_16x16x4), upper tile triangle only; DESIGN.md section 2.4" % (F + 1))
        mfma = {"achieved": tf, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MATRIX_PEAK_TFLOPS,
                "flops_per_launch": fl_launch, "depth": ld,
                "frac_at_measured_clock": (tf / (FP64_MATRIX_PEAK_TFLOPS * ghz / 2.4)) if ghz else None,
                "flops_per_evaluated_pair": fl_launch / evaluated_launch,
                # what HBM has to see at least: the feature matrix once, the partial sums of the depth splits once
                "algorithmic_bytes_per_launch": 8.0 * (N / n_gpus) * ld + 8.0 * evaluated_launch * depth_pieces(N, ld // 16),
                "other_kernels_in_step": "sig_features_kernel (the level features of every sequence), sig_gram_reduce_sym_kernel (adds the "
                                         "depth splits in a fixed order, mirrors); kernel_share_of_step says how much they and the launches cost",
                "kernel_share_of_step": per_launch_ms * launches_per_step / (dt / steps * 1e3)}
        tflops_exec = tf
        alu_peak = FP64_MATRIX_PEAK_TFLOPS
        alu_frac = tf / alu_peak
        f_exec = fl_launch / evaluated_launch
    if timed_kernel == "tvs_features_dgemm":      # the linear kernel's Kzx: one product of the tensors' and the sequences' level features
        fl_launch = mfma_flops / max(launches, 1)
        tf = fl_launch / (per_launch_ms * 1e-3) / 1e12
        bound = "mfma"
        binding = ("float64 matrix cores (rocBLAS dgemm): Kzx of the linear kernel is ONE product of the inducing tensors' rank-one level features "
                   "(T x ld) and the sequences' level features (N x ld), ld = sum_m d^m + 1 padded to 16; DESIGN.md section 6")
        mfma = {"achieved": tf, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MATRIX_PEAK_TFLOPS, "flops_per_launch": fl_launch,
                "frac_at_measured_clock": (tf / (FP64_MATRIX_PEAK_TFLOPS * ghz / 2.4)) if ghz else None,
                "other_kernels_in_step": "sig_features_kernel (the sequences' level features, weights and normalisation on them), "
                                         "tens_level_features_kernel, tens_gram_tile_kernel (Kzz)",
                "kernel_share_of_step": per_launch_ms * launches_per_step / (dt / steps * 1e3)}
        tflops_exec, alu_peak, alu_frac = tf, FP64_MATRIX_PEAK_TFLOPS, tf / FP64_MATRIX_PEAK_TFLOPS
        f_exec = fl_launch / evaluated_launch
        issue = None
    kernel_name = ("rocBLAS dgemm (product of level features)" if timed_kernel == "tvs_features_dgemm" else
                   timed_kernel + " (feature contraction on the float64 matrix cores)" if mfma else
                   "tvs_tile_kernel (tensor-vs-sequence chains)" if T else
                   ("seq_pk2_kernel (pair recursion, two sequences per pair group)" if (w["dtype"] == "f32" and base == "rbf")
                    else "seq_gram_kernel (pair recursion)"))
    # ---- HBM traffic: measured by this run (two rocprofv3 --pmc passes of the same command), else the committed passes
    tr, tr_src = None, None
    key = ("c2" if cfg == "c4" else cfg) + "_" + base + ("_increments" if increments else "") + ("_features" if mfma else "")
    if traffic == "measure" and n_gpus == 1 and cfg != "c4":
        m = measure_traffic(cfg, base, increments, lattice)
        if m:
            tr = m["bytes_per_launch"]
            tr_src = {"how": "measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each, of `bench.py --config %s --base %s%s "
                             "--steps 2 --warmup 1`; median over the dominant kernel's dispatches; FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KiB"
                             % (cfg, base, " --increments" if increments else ""), **m}
    if tr is None and traffic != "off":
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        try:
            ent = json.load(open(tf)).get(key) if cfg != "c4" else None
            if ent:
                tr = ent.get("bytes_per_launch")
                tr_src = {"how": "static: profiles/hbm_traffic.json (%s)" % ent.get("source", "round %s" % ent.get("round")),
                          "kernel": ent.get("kernel"), "grid": ent.get("grid")}
        except Exception:
            tr = None
    what = ("SVGP inducing-tensor path Kzz + Kzx + Kxx-diag (K_tens_n_seq_covs), T=%d inducing tensors%s, " % (T, " (increments)" if increments else "")
            if T else "full N x N Gram%s, " % (" sharded over %d GPUs, RCCL gather to rank 0" % n_gpus if n_gpus > 1 else ""))
    cname = "Signature" + ("Linear" if base == "linear" else "RBF")
    res = {
        "metric": METRIC,
        "value": value, "unit": "sequence-pairs/s", "n_gpus": n_gpus, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if (n_gpus == 1 or weak or cfg == "c2") else "strong", "vs_baseline": None,
        "dtype": w["dtype"], "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[{BASELINE_INDEX[cfg]}]: {what}{cname}, N={N}, L={L}, d={D}, num_levels={M}, "
                               f"order={int(order)}, normalization=on, {'fp64' if w['dtype'] == 'f64' else 'fp32'}, "
                               f"{'white-noise' if w['data'] == 'white' else 'random-walk'} inputs",
                   "name": cfg, "N": N, "L": L, "d": D, "num_levels": M, "order": int(order), "normalization": True,
                   "pairs_per_step": pairs,
                   "parallelism": ((f"sequence blocks x{n_gpus} (Z replicated), Kzx / Kxx-diag blocks gathered to rank 0 over RCCL" if T else
                                    f"owned-row blocks x{n_gpus}, {chunks} chunks per rank, compact (N/2+1 wide) rows gathered "
                                    f"asynchronously to rank 0 over RCCL, symmetrised there") if n_gpus > 1 else "single GPU")},
        "clock_ghz": ghz, "clock": clock,
        "roofline": {"bound": bound, "binding_limit": binding,
                     "issue_frac": issue["issue_frac"] if issue else None, "issue_model": issue,
                     "alu_frac": alu_frac,
                     # the contract's fields: ALGORITHMIC bytes per launch / the kernel's HIP-event time, against the HBM peak
                     "achieved": mfma["achieved"] if mfma else achieved, "peak": mfma["peak"] if mfma else HBM_PEAK_GBS,
                     "unit": mfma["unit"] if mfma else "GB/s",
                     "frac": mfma["frac"] if mfma else (stream_frac if stream_frac <= 1.0 else None), "stream_frac": stream_frac,
                     "mfma": mfma, "stream_achieved_gbs": achieved,
                     "frac_is": ("achieved / peak of the float64 matrix cores for the dominant kernel (`mfma`); stream_frac beside it is the ALGORITHMIC "
                                 "pair-stream rate of SURVEY 8(d) over the HBM peak, the number north_star's 60 % target is stated in -- not DRAM "
                                 "bandwidth (`traffic` is what the memory side saw)") if mfma else
                                "stream_frac: the ALGORITHMIC pair-stream rate of SURVEY 8(d) (every delivered entry priced at its input streams + its "
                                "result) over the HBM peak -- the number north_star's 60 % target is stated in -- NOT DRAM bandwidth: the streams are "
                                "served from L2 / LDS, `traffic` is what the memory side saw, `bound` names what limits the kernel"
                                + ("" if stream_frac <= 1.0 else "; above 1 here (a sequence record is staged once per 64 tensors), so it is not printed as a fraction of HBM"),
                     "traffic": tr, "traffic_source": tr_src,
                     # the same fraction with the kernel's duration as rocprofv3 recorded it in the traffic passes (mean over their dispatches):
                     # `frac` is from HIP events of un-profiled launches -- the two bracket what a reader of profiles/ will compute
                     "frac_profiled": ((mfma["flops_per_launch"] / (tr_src["kernel_us_profiled_mean"] * 1e-6) / 1e12 / FP64_MATRIX_PEAK_TFLOPS)
                                       if (mfma and tr_src and tr_src.get("kernel_us_profiled_mean")) else None),
                     "kernel": kernel_name, "kernel_ms_per_launch": per_launch_ms, "launches_per_step": launches_per_step,
                     "algorithmic_bytes_per_pair": b_pair, "pairs_per_launch": pairs_launch,
                     "alu": {"executed_flops_per_evaluated_pair": f_exec, "evaluated_pairs_per_launch": evaluated_launch,
                             "achieved_tflops": tflops_exec, "peak_tflops": alu_peak, "alu_frac": alu_frac,
                             "reference_flops_per_pair": f_ref,
                             "reference_flops_frac": (pairs_launch * f_ref / (per_launch_ms * 1e-3) / 1e12) / alu_peak}},
    }
    if rank_times and all(t is not None for t in rank_times):
        comp = [t["compute_ms"] for t in rank_times]
        res["per_rank"] = {"what": "the last timed step on every rank, ms of the stream it ran on: compute_ms = the row-block kernels of the rank's chunks, "
                                   "gather_inline_ms = what the stream spent starting the gathers between chunks (asynchronous under RCCL), gather_wait_ms = "
                                   "waiting for the gathers after the last chunk, symmetrise_ms = rank 0's pass from compact rows to the full matrix",
                           "ranks": rank_times, "compute_ms_max": max(comp), "compute_ms_min": min(comp),
                           "compute_max_over_min": (max(comp) / min(comp)) if min(comp) > 0 else None,
                           "gather_wait_ms_rank0": rank_times[0]["gather_wait_ms"], "symmetrise_ms_rank0": rank_times[0]["symmetrise_ms"]}
    if n_gpus > 1 and not T:
        # What the measured step should be if nothing but the decomposition's own terms bends the curve: every rank rebuilds the
        # full level features (not sharded: ~5 ms at N = 32,768, measured on one GPU), contracts its 1/n of the tile triangle, ships its compact rows
        # to rank 0 over its own xGMI link (all but the last chunk under the next chunk's compute), and rank 0 symmetrises.
        one_gpu_ms = 547.0 * (float(N) / 32768.0) ** 2          # c4-single-gpu of BENCH_r05 / profiles/r05_bench_default.json, scaled by pairs
        feat_ms = 5.0 * (float(N) / 32768.0)
        link_gbs = 120.0                                        # sustained per xGMI link (153 GB/s peak: /opt/skills/guides/MI355X_MICROARCH.md)
        rows_bytes = float(N) * (N // 2 + 1) * 8.0 / n_gpus     # one rank's compact rows
        last_gather_ms = rows_bytes / max(chunks, 1) / (link_gbs * 1e9) * 1e3
        sym_ms = (float(N) * N * 8.0 + float(N) * (N // 2 + 1) * 8.0) / 4.0e12 * 1e3
        pred = (one_gpu_ms - feat_ms) / n_gpus + feat_ms + last_gather_ms + sym_ms
        res["predicted"] = {"ms_per_step": pred, "speedup_ceiling": one_gpu_ms / pred,
                            "terms_ms": {"contraction_share": (one_gpu_ms - feat_ms) / n_gpus, "features_rebuilt_on_every_rank": feat_ms,
                                         "last_chunk_gather_exposed": last_gather_ms, "symmetrise_on_rank0": sym_ms},
                            "assumes": "one-GPU step %.0f ms (measured, round 5), %.0f GB/s per xGMI link, 4 TB/s for the symmetrisation pass; "
                                       "compare with ms_per_step and per_rank" % (one_gpu_ms, link_gbs)}
    if not T:
        res["config"]["route"] = ("explicit level features + one float64 matrix-core contraction (the linear base kernel has a finite feature space: "
                                  "K_m(x, y) = <Phi_m(x), Phi_m(y)>, same numbers as the pair recursion to rounding; the recursion itself is the "
                                  "`c2-linear-lattice` entry of `secondary`)" if mfma else
                                  "pair recursion: one sweep over the increment lattice per sequence pair, 16 lanes per pair")
        res["config"]["unique_pairs_computed_per_step"] = float(N) * (N + 1) / 2
        res["config"]["note"] = ("a step delivers all N*N Gram entries; symmetry is exploited on chip (each unordered pair is "
                                 "evaluated once and stored twice), as the reference's K(X) contract allows")
        res["roofline"]["stream_frac_on_unique_pairs"] = stream_frac * (N + 1) / (2.0 * N)
    if not checks:
        return res
    # ---- checks, outside the timed region -------------------------------------------------------------------
    res["rel_err"] = oracle_rel_err(cfg, w, base, increments, Xh, Zh, out)
    res["rel_err_note"] = ("max |K - K_oracle| / (|K_oracle| + 1e-6 max|K_oracle|) on a sub-sample of the timed output; "
                           "tolerance " + ("1e-6 (fp64)" if w["dtype"] == "f64" else "1e-4 (fp32 against the fp64 oracle)"))
    assert res["rel_err"] <= (1e-6 if w["dtype"] == "f64" else 1e-4), res["rel_err"]
    if not T:
        assert np.allclose(out[:8, :8].diagonal().cpu().numpy(), M + 1.0, atol=1e-9 if w["dtype"] == "f64" else 1e-4)
    if n_gpus > 1 and T:  # the gathered covariances against one single-context evaluation
        one = kern.K_tens_n_seq_covs(Z, X, increments=increments)
        res["verify_max_abs_diff_vs_single_rank"] = max(float((a - b).abs().max().item()) for a, b in zip(out, one))
    elif n_gpus > 1:      # the gathered Gram against single-context evaluations: leading block, and a block across rank boundaries
        nb = min(N, 1024)
        d1 = float((out[:nb, :nb] - kern.K(X[:nb])).abs().max().item())
        a, b = N - 300, N // 2 - 100
        d2 = float((out[a:a + 256, b:b + 256] - kern.K(X[a:a + 256], X[b:b + 256])).abs().max().item())
        res["verify_max_abs_diff_vs_single_rank"] = max(d1, d2)
    if host_e2e and n_gpus == 1 and cfg != "c4":      # the same evaluation from and to host memory (numpy in, numpy out)
        Xn = Xh.astype(np.float64 if w["dtype"] == "f64" else np.float32)
        Zn = Zh.astype(Xn.dtype) if T else None
        f = (lambda: kern.K_tens_n_seq_covs(Zn, Xn, increments=increments)) if T else (lambda: kern.K(Xn))
        f()
        t1 = time.perf_counter()
        for _ in range(3):
            f()
        res["end_to_end_ms_host_pointers"] = (time.perf_counter() - t1) / 3 * 1e3
    return res


SECONDARY = [("c2", "linear", False, True, 1), ("c2", "linear", False, False, 5), ("c2", "rbf", False, False, 1), ("c3", "rbf", False, False, 1),
             ("c3", "rbf", True, False, 1), ("c3", "linear", False, False, 1), ("c5", "rbf", False, False, 1)]
# round 6: the higher-order algorithms (signature_algs.py:37-74, 129-160) with SignatureRBF -- exact instances / higher-order chains in the tile kernel
# (after the other records: the tests address those by position)
SECONDARY_HO = [("c2", "rbf", False, False, 2), ("c3", "rbf", False, False, 2)]


def secondary_lines(dev):
    """The other single-GPU configurations, 3 warm-up + 10 steps each, as short records beside the headline."""
    out = []
    for cfg, base, inc, lattice, order in SECONDARY + SECONDARY_HO:
        name = cfg + "-" + base + ("-increments" if inc else "") + ("-lattice" if lattice else "") + ("-order%d" % order if order > 1 else "")
        try:
            r = run_workload(cfg, base, inc, 10, 3, dev, host_e2e=False, traffic="static", lattice=lattice, order=order)
            rf = r["roofline"]
            out.append({"name": name, "workload": r["config"]["workload"], "issue_source": (rf["issue_model"] or {}).get("source"), "dtype": r["dtype"], "value": r["value"], "unit": r["unit"],
                        "ms_per_step": r["ms_per_step"], "kernel_ms": rf["kernel_ms_per_launch"] * rf["launches_per_step"],
                        "kernel": rf["kernel"], "bound": rf["bound"], "alu_frac": rf["alu_frac"], "issue_frac": rf["issue_frac"],
                        "stream_frac": rf["stream_frac"], "clock_ghz": r["clock_ghz"],
                        "clock_ghz_xcd_min_max": ([min(q["ghz"] for q in r["clock"]["per_xcd"]), max(q["ghz"] for q in r["clock"]["per_xcd"])]
                                                  if r.get("clock") and r["clock"].get("per_xcd") else None),
                        "rel_err": r["rel_err"],
                        "traffic": rf["traffic"], "traffic_source": (rf["traffic_source"] or {}).get("how")})
        except Exception as e:                      # a side record never costs the headline its line
            out.append({"name": name, "error": repr(e)})
    ho = out[len(SECONDARY):]
    del out[len(SECONDARY):]
    out.append(c4_single_gpu_line(dev))
    out.extend(gradient_lines(dev))
    out.extend(svgp_lines(dev))
    out.extend(ho)
    return out


def svgp_lines(dev):
    """Round 6.  (i) One SVGP training step -- -ELBO forward + backward: covariances, conditional, KL, likelihood -- at the reference's OWN run settings
    (benchmarks/run_gpsig_benchmarks.py:32: 500 inducing tensors with increments, num_levels=4, num_lags=1, minibatch 50, SignatureRBF, time-augmented
    data: 2 (n_features + 1) columns) for four of its data sets' shapes (benchmarks/datasets.json; synthetic paths): 8, 10, 28 and 126 columns.  The
    last three run the wide route (csrc/wide_api.hip).  column_fma_per_s: tensors x sequences x observations x 20 component points x columns per second
    of the covariances' forward + backward, the yardstick that compares widths.  (ii) BASELINE configs[2] end to end: models.SVGP.predict_f --
    Kuu_Kuf_Kff (gpsig/inducing_variables.py:51-66) + Cholesky + two triangular solves + mean / variance (gpsig/models.py:62-73) -- at T = 512 inducing
    tensors, N = 16,384 sequences, with the split covariances / linear algebra."""
    import numpy as np
    import torch
    out = []
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    try:
        import reference_shapes as RS
    except Exception as e:              # noqa: BLE001
        return [{"name": "svgp-step", "error": repr(e)}]
    for ds in ("CharacterTrajectories", "NetFlow", "ArabicDigits", "CMUsubject16"):
        name = "svgp-step-" + ds.lower()
        try:
            r = RS.measure(ds, ["auto"], 5, device=str(dev), quiet=True)
            s, a = RS.shape_of(ds), r["auto"]
            work = float(s["T"]) * s["N"] * s["L"] * 20 * s["d_eff"]
            out.append({"name": name, "workload": "one SVGP step at the reference's settings for %s's shape: T=500 inducing tensors (increments), minibatch N=%d, "
                                                  "L=%d, %d columns (num_lags=1, time-augmented), num_levels=4, SignatureRBF, %d classes, fp64, synthetic paths"
                                                  % (ds, s["N"], s["L"], s["d_eff"], s["classes"]),
                        "dtype": "f64", "ms_per_step": a.get("step_ms"), "covs_fwd_ms": a.get("covs_fwd_ms"), "covs_fwd_bwd_ms": a.get("covs_fwd_bwd_ms"),
                        "kzx_fwd_bwd_ms": a.get("kzx_fwd_bwd_ms"), "kzz_fwd_bwd_ms": a.get("kzz_fwd_bwd_ms"), "kxx_diag_fwd_bwd_ms": a.get("kxx_diag_fwd_bwd_ms"),
                        "column_fma_per_s": (work / (a["covs_fwd_bwd_ms"] * 1e-3)) if a.get("covs_fwd_bwd_ms") else None, "error": a.get("error")})
        except Exception as e:          # noqa: BLE001
            out.append({"name": name, "error": repr(e)})
    out.append(c3_predict_line(dev))
    return out


def c3_predict_line(dev):
    """BASELINE configs[2] end to end (see svgp_lines)."""
    import numpy as np
    import torch
    out = []
    try:
        from gpsig_amd import inducing_variables as iv, kernels, models
        w = WORKLOADS["c3"]
        T, N, L, D, M = w["T"], w["N"], w["L"], w["d"], w["M"]
        rng = np.random.default_rng(0)
        X = torch.tensor(np.cumsum(rng.standard_normal((N, L, D)) * 0.3, axis=1).reshape(N, -1), device=dev)
        Z = torch.tensor(rng.standard_normal((M * (M + 1) // 2, T, 2, D)), device=dev)          # (inputs resident in HBM, as the headline's)
        kern = kernels.SignatureRBF(L * D, D, M, lengthscales=math.sqrt(D))
        q_sqrt = np.tile((0.5 * np.eye(T))[None], [1, 1, 1])
        m = models.SVGP(kern, iv.InducingTensors(Z, M, increments=True), num_latent=1, q_mu=rng.standard_normal((T, 1)), q_sqrt=q_sqrt, device=str(dev))

        def timed(fn, reps=5):
            fn(); torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / reps * 1e3
        covs_ms = timed(lambda: m._covs(X, False))
        total_ms = timed(lambda: m.predict_f(X))
        fm, fv = m.predict_f(X)
        out.append({"name": "c3-svgp-predict", "workload": "BASELINE.json configs[2] end to end: models.SVGP.predict_f (Kzz + Kzx + Kxx-diag, Cholesky, two triangular "
                                                           "solves, mean / variance), T=%d inducing tensors (increments), N=%d, L=%d, d=%d, num_levels=%d, SignatureRBF, "
                                                           "whitened, full q_sqrt, fp64" % (T, N, L, D, M),
                    "dtype": "f64", "ms_per_step": total_ms, "covariances_ms": covs_ms, "linear_algebra_ms": total_ms - covs_ms,
                    "value": float(T) * N / (total_ms * 1e-3), "unit": "(tensor, sequence) pairs/s, end to end",
                    "finite": bool(torch.isfinite(fm).all() and torch.isfinite(fv).all() and (fv > 0).all())})
    except Exception as e:              # noqa: BLE001
        out.append({"name": "c3-svgp-predict", "error": repr(e)})
    return out[0]


def c4_single_gpu_line(dev):
    """BASELINE configs[3] (N = 32,768) on ONE GPU: the same workload the N > 1 lines of a scaling series run, so that the series has an
    N = 1 point of its own (the driver's N = 1 line is configs[1]).  1 warm-up + 3 steps of ~0.55 s."""
    try:
        r = run_workload("c4", "linear", False, 3, 1, dev, host_e2e=False, traffic="off", issue_src="off")
        rf = r["roofline"]
        return {"name": "c4-single-gpu", "workload": r["config"]["workload"], "dtype": r["dtype"], "value": r["value"], "unit": r["unit"],
                "ms_per_step": r["ms_per_step"], "kernel_ms": rf["kernel_ms_per_launch"] * rf["launches_per_step"], "kernel": rf["kernel"],
                "bound": rf["bound"], "alu_frac": rf["alu_frac"], "stream_frac": rf["stream_frac"], "clock_ghz": r["clock_ghz"], "rel_err": r["rel_err"],
                "note": "the N = 1 point of the configs[3] scaling series: python bench.py --gpus N (N > 1) evaluates this very Gram sharded over N ranks"}
    except Exception as e:
        return {"name": "c4-single-gpu", "error": repr(e)}


def gradient_lines(dev):
    """Forward + backward of K(X) through gpsig_amd.autodiff (what the reference's TensorFlow autodiff does when it trains,
    training.py:149-164) at 1,024 sequences of BASELINE configs[1]'s shape: the linear kernel through the feature contraction's reverse
    pass (round 4: level sum, normalisation and weights inside one op, gpsig_kernel_K_grad; and through the level primitives with torch ops
    around them) and through the pair kernels' (option sig_features_grad = 0), and the RBF kernel.  Side records: ms per step only."""
    import numpy as np
    import torch
    from gpsig_amd import _lib, autodiff, kernels
    out = []
    N, L, D, M = 1024, 64, 8, 5
    rng = np.random.default_rng(0)
    X = torch.tensor(rng.standard_normal((N, L * D)), device=dev)
    W = torch.tensor(rng.standard_normal((N, N)), device=dev)
    ctx = _lib.context(dev.index or 0, torch.cuda.current_stream(dev).cuda_stream)
    # (name, class, option sig_features_grad, level sum and its gradient as one op -- gpsig_kernel_K_grad -- or level primitives + torch ops)
    # (name, class, option sig_features_grad, one op, (N, L): None = the headline shape's 1,024 x 64)
    for name, cls, opt, one_op, shape in (("grad-c2shape-n1024-linear", kernels.SignatureLinear, -1, True, None),
                                          ("grad-c2shape-n1024-linear-level-primitives", kernels.SignatureLinear, -1, False, None),
                                          ("grad-c2shape-n1024-linear-pair-kernels", kernels.SignatureLinear, 0, False, None),
                                          ("grad-c2shape-n1024-rbf", kernels.SignatureRBF, -1, True, None),
                                          # round 5's fused reverse kernel beyond its headline instance: another stationary family, and 128
                                          # observations per sequence (32 lanes per pair, two pairs per wavefront)
                                          ("grad-c2shape-n1024-matern32", kernels.SignatureMatern32, -1, True, None),
                                          ("grad-n512-l128-rbf", kernels.SignatureRBF, -1, True, (512, 128))):
        try:
            if shape is not None:
                N, L = shape
                X = torch.tensor(np.random.default_rng(0).standard_normal((N, L * D)), device=dev)
                W = torch.tensor(np.random.default_rng(1).standard_normal((N, N)), device=dev)
            kern = cls(L * D, D, M, lengthscales=(math.sqrt(D) if cls is not kernels.SignatureLinear else 1.0))
            mod = autodiff.SignatureKernelModule(kern, device=dev)
            mod.sum_route = one_op
            ctx.set_option("sig_features_grad", opt)

            def step():
                mod.zero_grad()
                (mod.K(X) * W).sum().backward()
            for _ in range(2):
                step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                step()
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) / reps * 1e3
            out.append({"name": name, "workload": "K(X) forward + backward through gpsig_amd.autodiff.SignatureKernelModule, N=%d, L=%d, d=%d, num_levels=%d, "
                                                  "normalization=on, fp64" % (N, L, D, M),
                        "dtype": "f64", "ms_per_step": ms, "value": float(N) * N / (ms * 1e-3), "unit": "sequence-pairs/s (forward + backward)"})
        except Exception as e:
            out.append({"name": name, "error": repr(e)})
        finally:
            ctx.set_option("sig_features_grad", -1)
    # round 5: the Matern families at compile time in the sequence Gram's evaluation kernel -- configs[1]'s Gram with SignatureMatern32 (time only)
    try:
        N, L = 4096, 64
        rng = np.random.default_rng(0)
        Xh = torch.tensor(np.cumsum(rng.standard_normal((N, L, D)) * 0.3, axis=1).reshape(N, -1), device=dev)
        kern = kernels.SignatureMatern32(L * D, D, M, lengthscales=math.sqrt(D))
        kern.K(Xh)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(3):
            kern.K(Xh)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / 3 * 1e3
        out.append({"name": "c2-matern32", "workload": "BASELINE.json configs[1] with SignatureMatern32: full N x N Gram, N=%d, L=%d, d=%d, num_levels=%d, "
                                                       "normalization=on, fp64, random-walk inputs (time only)" % (N, L, D, M),
                    "dtype": "f64", "ms_per_step": ms, "value": float(N) * (N + 1) / 2 / (ms * 1e-3), "unit": "sequence-pairs/s"})
    except Exception as e:
        out.append({"name": "c2-matern32", "error": repr(e)})
    # round 6: reverse passes at order 2 -- the sequence recursion's as two sweeps of a wavefront per pair (csrc/grad_wave_ho_kernel.hpp) inside the wide
    # route's dgemms, the chains' in the tile kernel (tvs_grad_tile_inst_ho.hip)
    try:
        N, L, M = 512, 64, 4
        D = 8
        Xg = torch.tensor(np.cumsum(np.random.default_rng(0).standard_normal((N, L, D)) * 0.3, axis=1).reshape(N, -1), device=dev)
        Wg = torch.tensor(np.random.default_rng(1).standard_normal((N, N)), device=dev)
        mod = autodiff.SignatureKernelModule(kernels.SignatureRBF(L * D, D, M, order=2, lengthscales=math.sqrt(D)), device=dev)

        def step2():
            mod.zero_grad()
            (mod.K(Xg) * Wg).sum().backward()
        for _ in range(2):
            step2()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(3):
            step2()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / 3 * 1e3
        out.append({"name": "grad-n512-rbf-order2", "workload": "K(X) forward + backward through gpsig_amd.autodiff.SignatureKernelModule, SignatureRBF order=2, N=%d, L=%d, d=%d, "
                                                                "num_levels=%d, normalization=on, fp64" % (N, L, D, M),
                    "dtype": "f64", "ms_per_step": ms, "value": float(N) * N / (ms * 1e-3), "unit": "sequence-pairs/s (forward + backward)"})
    except Exception as e:
        out.append({"name": "grad-n512-rbf-order2", "error": repr(e)})
    try:
        T, N, L, d3, M = 512, 16384, 50, 6, 4
        rng = np.random.default_rng(0)
        X3 = torch.tensor(rng.standard_normal((N, L * d3)) * 0.3, device=dev)
        Z3 = torch.tensor(rng.standard_normal((M * (M + 1) // 2, T, d3)) * 0.3, device=dev).requires_grad_(True)
        mod = autodiff.SignatureKernelModule(kernels.SignatureRBF(L * d3, d3, M, order=2), device=dev)

        def step3():
            Z3.grad = None
            mod.zero_grad()
            o = mod.K_tens_vs_seq(Z3, X3)
            (o * o).sum().backward()
        for _ in range(2):
            step3()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(3):
            step3()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / 3 * 1e3
        out.append({"name": "grad-c3-rbf-order2", "workload": "Kzx forward + backward through gpsig_amd.autodiff.SignatureKernelModule at BASELINE configs[2]'s size, SignatureRBF order=2: "
                                                              "T=%d inducing tensors, N=%d, L=%d, d=%d, num_levels=%d, normalization=on, fp64" % (T, N, L, d3, M),
                    "dtype": "f64", "ms_per_step": ms, "value": float(T) * N / (ms * 1e-3), "unit": "tensor-sequence pairs/s (forward + backward)"})
    except Exception as e:
        out.append({"name": "grad-c3-rbf-order2", "error": repr(e)})
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--base", default=None, choices=["linear", "rbf"])
    ap.add_argument("--increments", action="store_true", help="c3: inducing tensors hold increments (kernels.py:329-330)")
    ap.add_argument("--weak", action="store_true", help="--gpus N > 1: N_total = 4096 * sqrt(N) instead of configs[3]")
    ap.add_argument("--chunks", type=int, default=4, help="pieces a rank's row block is computed / gathered in")
    ap.add_argument("--lattice", action="store_true", help="SignatureLinear through the pair recursion instead of the feature contraction")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="default line only: leave out the other single-GPU configurations")
    ap.add_argument("--no-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes (roofline.traffic falls back to profiles/hbm_traffic.json)")
    ap.add_argument("--timed-loop-only", action="store_true", help="warm-up + timed steps and nothing else (what the --pmc passes run)")
    ap.add_argument("--rendezvous-check", action="store_true", help="start the ranks, gather ranks_seen, print it; no evaluation (CPU-testable)")
    args = ap.parse_args(argv)
    argv = list(sys.argv[1:] if argv is None else argv)

    backend = os.environ.get("GPSIG_BENCH_BACKEND", "nccl")       # "gloo": functional check of the N > 1 path on a box with fewer GPUs
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    in_torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not in_torchrun:
        # plain `python bench.py --gpus N`: be the launcher.  Never fall back to fewer ranks than asked for.
        if backend == "nccl" and not args.rendezvous_check and visible_gpus() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: this node shows {visible_gpus()} GPU(s) to HIP; one rank per GPU over RCCL needs {args.gpus} "
                             "(GPSIG_BENCH_BACKEND=gloo runs the ranks on the GPUs there are, as a functional check)")
        raise SystemExit(launch_ranks(args.gpus, argv))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: a line must say how many ranks computed it")
    n_gpus = world
    cfg = args.config or ("c2" if n_gpus == 1 else "c4")
    if n_gpus > 1 and cfg not in ("c2", "c3", "c4"):
        raise SystemExit("--gpus N > 1 runs the sharded symmetric Gram (c4, or c2 with --weak) or the sequence-sharded SVGP covariances (c3); "
                         "c5 is a single-GPU workload")
    base = args.base or WORKLOADS[cfg]["base"]

    cpu = None
    if rank == 0 and not (args.no_cpu_baseline or args.timed_loop_only or args.rendezvous_check):
        # before CUDA is initialised in this process; the N > 1 lines carry the C restatement only (the other ranks wait in the rendezvous)
        cpu = cpu_baseline(cfg, base, args.increments, numpy_leg=(n_gpus == 1))

    import torch
    import torch.distributed as dist

    ndev = visible_gpus()
    if args.rendezvous_check:
        dev_index = (local_rank % ndev) if ndev else None
    else:
        if backend == "nccl" and ndev < max(n_gpus, 1):
            raise SystemExit(f"rank {rank}: {ndev} GPU(s) visible, {n_gpus} ranks over RCCL need one each")
        if ndev < 1:
            raise SystemExit("no GPU visible: gpsig_amd has no CPU path")
        dev_index = local_rank if backend == "nccl" else local_rank % ndev
    dev = None
    if dev_index is not None:
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    ranks_seen = [rank_identity(rank, local_rank, dev_index, backend if world > 1 else None)]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != n_gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {n_gpus}")
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks_seen[0])
        ranks_seen = gathered
        if backend == "nccl" and len({(r["device"], r["pci_bus_id"]) for r in ranks_seen}) != world:
            raise SystemExit(f"ranks share a GPU under the nccl backend: {ranks_seen}")
    seen = {"world_size": dist.get_world_size() if world > 1 else 1, "backend": (dist.get_backend() if world > 1 else None),
            "launched_by": "bench.py itself" if os.environ.get("GPSIG_BENCH_LAUNCHED") else ("torch.distributed.run" if in_torchrun else "single process"),
            "ranks": ranks_seen}
    if args.rendezvous_check:
        if rank == 0:
            print(json.dumps({"n_gpus": n_gpus, "rendezvous_check": True, "ranks_seen": seen}))
        if world > 1:
            dist.destroy_process_group()
        return

    lean = args.timed_loop_only
    res = run_workload(cfg, base, args.increments, args.steps, args.warmup, dev, rank, world, chunks=args.chunks, weak=args.weak,
                       checks=not lean, host_e2e=not lean, lattice=args.lattice,
                       traffic="off" if lean else ("static" if (args.no_traffic or args.lattice) else "measure"),
                       issue_src="off" if (lean or args.no_traffic) else "measure")
    if rank == 0:
        res["ranks_seen"] = seen
        if cpu is not None:
            res["cpu_baseline"] = cpu
        if n_gpus == 1 and args.config is None and args.base is None and not args.increments and not args.no_secondary and not lean and not args.lattice:
            res["secondary"] = secondary_lines(dev)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
