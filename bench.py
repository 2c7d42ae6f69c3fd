#!/usr/bin/env python3
"""Headline benchmark: sequence-pairs/s of SignatureKernel.K (full N x N Gram) on MI355X.

Workload (BASELINE.json configs[1]): N=4096 sequences, L=64 observations, d=8 features, num_levels=5,
fp64, SignatureLinear (the esig-validated kernel class of notebooks/signature_kernel.ipynb), order 1,
level normalisation on, white-noise inputs (notebook cell 4).  One step = one complete kern.K(X) with X
already resident in HBM and the (N, N) result left in HBM.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1 (weak scaling): the Gram of N_total = 4096 * sqrt(N) sequences is cut into independent pair blocks,
every rank computes its share (gpsig_set_shard) into a compact buffer and rank 0 gathers them over RCCL.
Prints ONE JSON line on rank 0.
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

N_BASE, L, D, M = 4096, 64, 8, 5
B_PAIR = (L + L) * D * 8 + 8                      # pair-stream bytes: both L x d streams + one fp64 result
F_PAIR = 2 * L * L * D + (L - 1) * (L - 1) * 4 * M   # reference op count per pair
F_EXEC = (L - 1) * (L - 1) * (2 * D + 3 * M - 1)      # fp64 flops the row-sweep kernel executes per evaluated pair
HBM_PEAK_GBS = 8000.0                             # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6


def cpu_tile(args):
    """One tile of the op-for-op CPU restatement (oracle) -- runs in a worker process."""
    import numpy as _np
    from oracle import sigkern_oracle as O
    seed, i0, j0, tile, n = args
    rng = _np.random.default_rng(seed)
    X = rng.standard_normal((n, L, D))
    kern = O.SignatureKernelOracle(L * D, D, M, base="linear", normalization=False, lengthscales=None)
    t0 = time.perf_counter()
    kern._K_seq(X[i0:i0 + tile], X[j0:j0 + tile])
    return time.perf_counter() - t0


def cpu_baseline(budget_s=15.0):
    """The reference's TF-CPU graph restated op for op in NumPy (oracle/, kind 'port'), on tiles of 32 x 32
    sequences at the benchmark shape, spread over worker processes.  Bounded sample; see DESIGN.md."""
    import multiprocessing as mp
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    workers = max(1, min(64, (os.cpu_count() or 2) // 2))
    tile = 32
    ctx = mp.get_context("spawn")
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    with ctx.Pool(workers) as pool:
        t1 = pool.map(cpu_tile, [(0, 0, 0, tile, tile)] * workers)          # warm-up + calibration
        per_tile = float(np.median(t1))
        rounds = max(1, int(budget_s / max(per_tile, 1e-3)))
        ntiles = workers * rounds
        t0 = time.perf_counter()
        pool.map(cpu_tile, [(0, 0, 0, tile, tile)] * ntiles)
        wall = time.perf_counter() - t0
    pairs = ntiles * tile * tile
    return {"value": pairs / wall, "unit": "sequence-pairs/s", "cores": workers, "kind": "port",
            "sample": f"{ntiles} tiles of {tile}x{tile} sequence pairs at L={L}, d={D}, num_levels={M}, fp64 "
                      f"(unnormalised levels: matmul, 4-slice difference, 2 cumsums + multiply + reduce per level; "
                      f"gpsig/kernels.py:226, gpsig/signature_algs.py:25-35), {workers} single-threaded NumPy "
                      f"worker processes, {wall:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--base", default="linear", choices=["linear", "rbf"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = max(world, 1)
    if args.gpus != n_gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()            # before CUDA is initialised in this process

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

    n_total = N_BASE if n_gpus == 1 else int(round(N_BASE * math.sqrt(n_gpus) / 64.0)) * 64
    rng = np.random.default_rng(0)
    X = torch.as_tensor(rng.standard_normal((n_total, L * D)), device=dev)      # same data on every rank
    cls = kernels.SignatureLinear if args.base == "linear" else kernels.SignatureRBF
    kern = cls(L * D, D, M, lengthscales=(1.0 if args.base == "linear" else math.sqrt(D)))
    gram = parallel.ShardedGram(kern, n_total, dev, rank, world)

    def step():
        return gram(X)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
    ctx = gram.ctx
    barrier()
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches, pairs_done = ctx.timing_get()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        entries = float(n_total) * n_total                       # Gram entries delivered per step
        value = entries * args.steps / dt
        per_launch_ms = kernel_ms / max(launches, 1)
        entries_rank = entries / n_gpus
        achieved = entries_rank * B_PAIR / (per_launch_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(args.base, {}).get("bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "sequence-pairs/sec for SignatureKernel.K (N,L,d,sig_level); fp64 rel-err vs ref",
            "value": value, "unit": "sequence-pairs/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[1]: full N x N Gram, Signature{args.base.upper() if args.base == 'rbf' else 'Linear'}.K, "
                                   f"N={n_total}, L={L}, d={D}, num_levels={M}, order=1, normalization=on, fp64, white-noise inputs",
                       "N": n_total, "L": L, "d": D, "num_levels": M, "order": 1, "normalization": True,
                       "pairs_per_step": entries, "unique_pairs_computed_per_step": float(n_total) * (n_total + 1) / 2,
                       "note": "a step delivers all N*N Gram entries; symmetry is exploited on chip (each unordered pair is "
                               "evaluated once and stored twice), as the reference's K(X) contract allows",
                       "parallelism": f"pair-block shards x{n_gpus}" + (", RCCL gather of compact shards to rank 0" if n_gpus > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "kernel": "seq_gram_kernel (pair recursion)", "kernel_ms_per_launch": per_launch_ms,
                         "algorithmic_bytes_per_pair": B_PAIR, "pairs_per_launch": entries_rank,
                         "alu_frac_fp64_vector": ((float(n_total) * (n_total + 1) / 2 / n_gpus) * F_EXEC / (per_launch_ms * 1e-3) / 1e12)
                                                 / FP64_VECTOR_PEAK_TFLOPS,
                         "reference_flops_frac_fp64_vector": (entries_rank * F_PAIR / (per_launch_ms * 1e-3) / 1e12) / FP64_VECTOR_PEAK_TFLOPS,
                         "frac_on_unique_pairs": achieved / HBM_PEAK_GBS * (n_total + 1) / (2.0 * n_total)},
        }
        if cpu is not None:
            res["cpu_baseline"] = cpu
        chk = out[:8, :8].diagonal().cpu().numpy() if out is not None else None
        if chk is not None:
            assert np.allclose(chk, M + 1.0, atol=1e-9), chk
        if os.environ.get("GPSIG_BENCH_VERIFY") and n_gpus > 1:      # the gathered Gram against the single-context one
            ref = kern.K(X)
            res["verify_max_abs_diff_vs_single_rank"] = float((out - ref).abs().max().item())
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
