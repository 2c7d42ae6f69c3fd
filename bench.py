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
on rank 0: the driver's contract fields + `roofline` (pair-stream fraction AND executed-flop ALU fraction of the dominant
kernel, timed with HIP events on the library's stream), `cpu_baseline` (the oracle's op-for-op restatement of the reference's
TF graph on the host cores, bounded sample), `rel_err` (a sub-sample of the timed output against the oracle, outside the timed
region) and `end_to_end_ms_host_pointers` (the same evaluation from and to host memory: H2D + compute + D2H).
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


def cpu_baseline(cfg, base, increments, budget_s=10.0):
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
    ko = O.SignatureKernelOracle(w["L"] * w["d"], w["d"], w["M"], base=base, lengthscales=lengthscales(dict(w, base=base)))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--base", default=None, choices=["linear", "rbf"])
    ap.add_argument("--increments", action="store_true", help="c3: inducing tensors hold increments (kernels.py:329-330)")
    ap.add_argument("--weak", action="store_true", help="--gpus N > 1: N_total = 4096 * sqrt(N) instead of configs[3]")
    ap.add_argument("--chunks", type=int, default=4, help="pieces a rank's row block is computed / gathered in")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = max(world, 1)
    if args.gpus != n_gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    cfg = args.config or ("c2" if n_gpus == 1 else "c4")
    if n_gpus > 1 and cfg not in ("c2", "c3", "c4"):
        raise SystemExit("--gpus N > 1 runs the sharded symmetric Gram (c4, or c2 with --weak) or the sequence-sharded SVGP covariances (c3); "
                         "c5 is a single-GPU workload")
    w = dict(WORKLOADS[cfg])
    base = args.base or w["base"]
    w["base"] = base
    if n_gpus > 1 and (args.weak or cfg == "c2"):
        w["N"] = int(round(4096 * math.sqrt(n_gpus) / 64.0)) * 64

    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, base, args.increments)      # before CUDA is initialised in this process

    import torch
    import torch.distributed as dist
    from gpsig_amd import _lib, kernels, parallel

    backend = os.environ.get("GPSIG_BENCH_BACKEND", "nccl")       # "gloo": functional check of the N > 1 path on a box with one GPU
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    N, L, D, M, T = w["N"], w["L"], w["d"], w["M"], w["T"]
    tdt = torch.float64 if w["dtype"] == "f64" else torch.float32
    Xh = make_inputs(w)                                              # same data on every rank
    X = torch.as_tensor(Xh, device=dev).to(tdt)
    Zh = make_tensors(w, args.increments) if T else None
    Z = torch.as_tensor(Zh, device=dev).to(tdt) if T else None
    cls = kernels.SignatureLinear if base == "linear" else kernels.SignatureRBF
    kern = cls(L * D, D, M, lengthscales=lengthscales(w))
    gram = parallel.ShardedGram(kern, N, dev, rank, world, chunks=args.chunks) if not T else None
    covs = parallel.ShardedCovs(kern, N, dev, rank, world) if T else None      # world == 1: kern.K_tens_n_seq_covs itself

    def step():
        if T:
            return covs(Z, X, increments=args.increments)
        return gram(X)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ctx = _lib.context(dev.index or 0, torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing_reset()
    t0 = time.perf_counter()
    out = None
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches, _ = ctx.timing_get()      # HIP events around the dominant kernel, on the stream it was launched on

    if world > 1:
        tt = torch.tensor([dt, kernel_ms / max(launches, 1)], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, per_launch_ms = float(tt[0].item()), float(tt[1].item())
        launches_per_step = launches / max(args.steps, 1)
    else:
        per_launch_ms = kernel_ms / max(launches, 1)
        launches_per_step = launches / max(args.steps, 1)

    if rank == 0:
        pairs = float(T) * N if T else float(N) * N                    # entries delivered per step
        value = pairs * args.steps / dt
        b_pair = stream_bytes_per_pair(w, args.increments)
        f_ref, f_exec = flops_per_pair(w, args.increments)
        # one launch of the dominant kernel: this rank's share of a step, divided over the launches it took
        pairs_launch = pairs / n_gpus / max(launches_per_step, 1)
        evaluated_launch = pairs_launch if T else pairs_launch * (N + 1) / (2.0 * N)     # symmetric Gram: each unordered pair once
        achieved = pairs_launch * b_pair / (per_launch_ms * 1e-3) / 1e9
        alu_peak = FP64_VECTOR_PEAK_TFLOPS if w["dtype"] == "f64" else FP32_VECTOR_PEAK_TFLOPS
        tflops_exec = evaluated_launch * f_exec / (per_launch_ms * 1e-3) / 1e12
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                key = ("c2" if cfg == "c4" else cfg) + "_" + base + ("_increments" if args.increments else "")
                ent = json.load(open(tf)).get(key) if cfg != "c4" else None
                if ent:
                    traffic = ent.get("bytes_per_launch")
                    traffic_src = "static: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, %s)" % (
                        "profiles/hbm_traffic.json", ent.get("source", "round %s" % ent.get("round")))
            except Exception:
                traffic = None
        kernel_name = ("tvs_tile_kernel (tensor-vs-sequence chains)" if T else
                       ("seq_pk2_kernel (pair recursion, two sequences per pair group)" if (w["dtype"] == "f32" and base == "rbf")
                        else "seq_gram_kernel (pair recursion)"))
        what = ("SVGP inducing-tensor path Kzz + Kzx + Kxx-diag (K_tens_n_seq_covs), T=%d inducing tensors%s, " % (T, " (increments)" if args.increments else "")
                if T else "full N x N Gram%s, " % (" sharded over %d GPUs, RCCL gather to rank 0" % n_gpus if n_gpus > 1 else ""))
        cname = "Signature" + ("Linear" if base == "linear" else "RBF")
        res = {
            "metric": METRIC,
            "value": value, "unit": "sequence-pairs/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if (n_gpus == 1 or args.weak or cfg == "c2") else "strong", "vs_baseline": None,
            "dtype": w["dtype"], "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{BASELINE_INDEX[cfg]}]: {what}{cname}, N={N}, L={L}, d={D}, num_levels={M}, "
                                   f"order=1, normalization=on, {'fp64' if w['dtype'] == 'f64' else 'fp32'}, "
                                   f"{'white-noise' if w['data'] == 'white' else 'random-walk'} inputs",
                       "name": cfg, "N": N, "L": L, "d": D, "num_levels": M, "order": 1, "normalization": True,
                       "pairs_per_step": pairs,
                       "parallelism": ((f"sequence blocks x{n_gpus} (Z replicated), Kzx / Kxx-diag blocks gathered to rank 0 over RCCL" if T else
                                        f"owned-row blocks x{n_gpus}, {args.chunks} chunks per rank, compact (N/2+1 wide) rows gathered "
                                        f"asynchronously to rank 0 over RCCL, symmetrised there") if n_gpus > 1 else "single GPU")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name, "kernel_ms_per_launch": per_launch_ms, "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_pair": b_pair, "pairs_per_launch": pairs_launch,
                         "binding_limit": ("float64 vector-ALU issue (the kernel runs within a few per cent of instructions x 4 cycles; "
                                           "DESIGN.md section 4)" if w["dtype"] == "f64" else
                                           "float32 dependent-instruction latency at 3 wavefronts per SIMD (DESIGN.md section 2.4)"),
                         "note": "frac is the ALGORITHMIC pair-stream rate of SURVEY 8(d) (every delivered entry priced at its two "
                                 "input streams + its result) over the HBM peak, NOT measured DRAM bandwidth: the streams are served "
                                 "from L2 / LDS, `traffic` is what HBM saw.  The binding limit of the pair recursion is vector-ALU "
                                 "issue: see alu_frac.",
                         "alu": {"executed_flops_per_evaluated_pair": f_exec, "evaluated_pairs_per_launch": evaluated_launch,
                                 "achieved_tflops": tflops_exec, "peak_tflops": alu_peak, "alu_frac": tflops_exec / alu_peak,
                                 "reference_flops_per_pair": f_ref,
                                 "reference_flops_frac": (pairs_launch * f_ref / (per_launch_ms * 1e-3) / 1e12) / alu_peak}},
        }
        if not T:
            res["config"]["unique_pairs_computed_per_step"] = float(N) * (N + 1) / 2
            res["config"]["note"] = ("a step delivers all N*N Gram entries; symmetry is exploited on chip (each unordered pair is "
                                     "evaluated once and stored twice), as the reference's K(X) contract allows")
            res["roofline"]["frac_on_unique_pairs"] = achieved / HBM_PEAK_GBS * (N + 1) / (2.0 * N)
        if cpu is not None:
            res["cpu_baseline"] = cpu
        # ---- checks, outside the timed region -------------------------------------------------------------------
        res["rel_err"] = oracle_rel_err(cfg, w, base, args.increments, Xh, Zh, out)
        res["rel_err_note"] = ("max |K - K_oracle| / (|K_oracle| + 1e-6 max|K_oracle|) on a sub-sample of the timed output; "
                               "tolerance " + ("1e-6 (fp64)" if w["dtype"] == "f64" else "1e-4 (fp32 against the fp64 oracle)"))
        assert res["rel_err"] <= (1e-6 if w["dtype"] == "f64" else 1e-4), res["rel_err"]
        if not T:
            assert np.allclose(out[:8, :8].diagonal().cpu().numpy(), M + 1.0, atol=1e-9 if w["dtype"] == "f64" else 1e-4)
        if n_gpus > 1 and T:  # the gathered covariances against one single-context evaluation
            one = kern.K_tens_n_seq_covs(Z, X, increments=args.increments)
            res["verify_max_abs_diff_vs_single_rank"] = max(float((a - b).abs().max().item()) for a, b in zip(out, one))
        elif n_gpus > 1:      # the gathered Gram against single-context evaluations: leading block, and a block across rank boundaries
            nb = min(N, 1024)
            d1 = float((out[:nb, :nb] - kern.K(X[:nb])).abs().max().item())
            a, b = N - 300, N // 2 - 100
            d2 = float((out[a:a + 256, b:b + 256] - kern.K(X[a:a + 256], X[b:b + 256])).abs().max().item())
            res["verify_max_abs_diff_vs_single_rank"] = max(d1, d2)
        if n_gpus == 1 and cfg != "c4":      # the same evaluation from and to host memory (numpy in, numpy out)
            Xn = Xh.astype(np.float64 if w["dtype"] == "f64" else np.float32)
            Zn = Zh.astype(Xn.dtype) if T else None
            f = (lambda: kern.K_tens_n_seq_covs(Zn, Xn, increments=args.increments)) if T else (lambda: kern.K(Xn))
            f()
            t1 = time.perf_counter()
            for _ in range(3):
                f()
            res["end_to_end_ms_host_pointers"] = (time.perf_counter() - t1) / 3 * 1e3
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
