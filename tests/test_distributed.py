"""The N > 1 path on CPU: two processes over the gloo backend run gpsig_amd.parallel.ShardedGram.__call__ ITSELF -- row
partition, chunked compact row blocks, asynchronous gathers, symmetrisation -- with the library context replaced by a stand-in
that executes the two C-ABI calls with the kernel's lock-step CPU emulator (tests/emu) and the NumPy statement of the
symmetrisation.  The same partition, ownership rule, task lists and epilogue the 8-GPU run uses."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from gpsig_amd import parallel


def test_row_partition_is_aligned_and_complete():
    for n in (1, 4, 7, 64, 100, 4096, 11584, 32768):
        for world in (1, 2, 3, 8):
            for align in (4, 16):
                b = parallel.row_partition(n, world, align)
                assert b[0] == 0 and b[-1] == n and len(b) == world + 1
                assert all(b[r] <= b[r + 1] for r in range(world))
                assert all(b[r] % align == 0 or b[r] == b[r + 1] for r in range(world))   # aligned, or an empty block


def test_ownership_rule_covers_each_unordered_pair_once():
    for n in (1, 2, 5, 8, 9):
        _, owned = parallel.symmetrize_reference(np.zeros((n, n)))
        off = ~np.eye(n, dtype=bool)
        assert (owned ^ owned.T)[off].all() and owned.diagonal().all()


def test_compact_layout_round_trip():
    """Packing the owned entries of a symmetric matrix row by row and unpacking them gives the matrix back."""
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 12, 65):
        A = rng.standard_normal((n, n))
        A = A + A.T
        h = n // 2
        owned = parallel.owned_mask(n)
        half = np.full((n, h + 1), np.nan)
        for r in range(n):
            for c in range(n):
                if owned[r, c]:
                    half[r, h - (r - c) % n] = A[r, c]
        assert np.isnan(half).sum() == (n // 2 if n % 2 == 0 else 0)   # even n: the tie at distance n/2 goes to one row of the pair
        np.testing.assert_array_equal(parallel.symmetrize_compact_reference(np.nan_to_num(half)), A)


class EmulatorContext:
    """Stand-in for gpsig_amd._lib.Context on CPU tensors: the two calls ShardedGram makes, executed by the emulator."""
    BASE = {0: "linear", 1: "rbf"}

    def call(self, name, p, Xp, n, L, r0, r1, outp):
        import emu_util as E
        assert name == "gpsig_kernel_K_symm_rows_compact"
        d, M = p.num_features, p.num_levels
        X = np.ctypeslib.as_array(C.cast(Xp, C.POINTER(C.c_double)), shape=(n, L, d))
        var = np.ctypeslib.as_array(p.variances, shape=(M + 1,))
        assert not p.lengthscales and p.num_lags == 0 and p.order == 1
        blk = E.kernel_K_owned_rows(X, self.BASE[p.base_kernel], M, var, p.sigma, bool(p.normalization), r0, r1, jitter=p.jitter,
                                    compact=True)
        out = np.ctypeslib.as_array(C.cast(outp, C.POINTER(C.c_double)), shape=(r1 - r0, n // 2 + 1))
        np.copyto(out, blk, where=blk != 0)            # "entries it does not own are left untouched"

    def symmetrize_compact_rows(self, dtype_id, halfp, n, outp):
        half = np.ctypeslib.as_array(C.cast(halfp, C.POINTER(C.c_double)), shape=(n, n // 2 + 1))
        out = np.ctypeslib.as_array(C.cast(outp, C.POINTER(C.c_double)), shape=(n, n))
        out[...] = parallel.symmetrize_compact_reference(half)


def _spawn_with_retry(worker, world, args_after_port, attempts=3):
    """Run `worker(rank, world, port, *args_after_port, ret)` on `world` gloo ranks.  The free port is found by binding and
    releasing it, which another process can win in between (rendezvous then fails): a transient of the test harness, not of the
    code under test -- retried on a fresh port."""
    last = None
    for _ in range(attempts):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        with mp.Manager() as mgr:
            ret = mgr.dict()
            try:
                mp.spawn(worker, args=(world, port) + tuple(args_after_port) + (ret,), nprocs=world, join=True)
                return dict(ret)
            except Exception as e:        # noqa: BLE001
                last = e
    raise last


def _worker(rank, world, port, n, chunks, ret):
    import torch
    from gpsig_amd import kernels
    from oracle import sigkern_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, d, M = 9, 3, 3
        rng = np.random.default_rng(0)
        X = np.cumsum(0.3 * rng.standard_normal((n, L, d)), axis=1).reshape(n, L * d)       # replicated input
        var = np.array([0.7, 1.1, 0.9, 1.3])
        kern = kernels.SignatureRBF(L * d, d, M, variances=var, lengthscales=None)
        gram = parallel.ShardedGram(kern, n, torch.device("cpu"), rank, world, chunks=chunks, ctx=EmulatorContext())
        out = gram(torch.from_numpy(X))
        ret["timings%d" % rank] = gram.timings()          # where this rank's time went (bench.py's per_rank record of an N > 1 line)
        if rank == 0:
            want = O.SignatureKernelOracle(L * d, d, M, base="rbf", variances=var, lengthscales=None).K(X)
            got = out.numpy()
            ret["err"] = float(np.abs(got - want).max() / np.abs(want).max())
            ret["sym"] = bool((got == got.T).all())
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,chunks", [(22, 1), (37, 2), (64, 4)])
def test_two_rank_gloo_sharded_gram(n, chunks):
    ret = _spawn_with_retry(_worker, 2, (n, chunks))
    assert ret["err"] < 1e-12 and ret["sym"]
    for r in (0, 1):
        t = ret["timings%d" % r]
        assert t["rank"] == r and t["chunks"] == chunks and t["compute_ms"] > 0 and t["gather_wait_ms"] >= 0 and t["gather_inline_ms"] >= 0
        assert (t["symmetrise_ms"] > 0) == (r == 0)
    assert ret["timings0"]["rows"] + ret["timings1"]["rows"] == n


class RefusingContext(EmulatorContext):
    """The row-block entry point refusing the shape, as libgpsig_hip does for shapes outside the wavefront kernels."""

    def call(self, name, p, Xp, n, L, r0, r1, outp):
        raise NotImplementedError("no sequence-pair kernel shape for these lengths")


def _fallback_worker(rank, world, port, n, ret):
    import torch
    from gpsig_amd import kernels
    from oracle import sigkern_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, d, M = 7, 2, 3
        rng = np.random.default_rng(4)
        X = np.cumsum(0.3 * rng.standard_normal((n, L, d)), axis=1).reshape(n, L * d)
        kern = kernels.SignatureRBF(L * d, d, M, lengthscales=None)
        ko = O.SignatureKernelOracle(L * d, d, M, base="rbf", lengthscales=None)
        calls = []

        def K_stand_in(Xt, presliced=False):        # stand-in for the any-shape HIP evaluation (no CPU path in the product)
            calls.append(rank)
            return torch.from_numpy(ko.K(Xt.numpy()))
        kern.K = K_stand_in
        gram = parallel.ShardedGram(kern, n, torch.device("cpu"), rank, world, chunks=2, ctx=RefusingContext())
        out = gram(torch.from_numpy(X))
        ret[rank] = (gram.fallback is not None, len(calls), None if out is None else float(np.abs(out.numpy() - ko.K(X)).max()))
    finally:
        dist.destroy_process_group()


def test_sharded_gram_falls_back_to_rank_zero_for_unsupported_shapes():
    """A shape the row-block kernels refuse: every rank takes the fallback together (also one that owns no rows: n = 3 on 3 ranks of
    blocks of 8), rank 0 evaluates alone through kern.K, nobody hangs in a gather."""
    for n, world in ((20, 2), (3, 3)):
        ret = _spawn_with_retry(_fallback_worker, world, (n,))
        assert ret[0] == (True, 1, 0.0), ret
        assert all(ret[r] == (True, 0, None) for r in range(1, world)), ret


class RefusingOnOneRank(EmulatorContext):
    """Only ONE rank's library refuses (a verdict that depends on the rank: an empty block planned by other rules than a block with
    rows, a device without memory for one route) -- the divergence that would leave its peers in a gather it never joins."""

    def __init__(self, refuse):
        self.refuse = refuse

    def call(self, name, p, Xp, n, L, r0, r1, outp):
        if self.refuse:
            raise NotImplementedError("no sequence-pair kernel shape for these lengths")
        if r1 > r0:
            EmulatorContext.call(self, name, p, Xp, n, L, r0, r1, outp)


def _diverging_worker(rank, world, port, n, refusing_rank, ret):
    import torch
    from gpsig_amd import kernels
    from oracle import sigkern_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, d, M = 7, 2, 3
        rng = np.random.default_rng(5)
        X = np.cumsum(0.3 * rng.standard_normal((n, L, d)), axis=1).reshape(n, L * d)
        kern = kernels.SignatureRBF(L * d, d, M, lengthscales=None)
        ko = O.SignatureKernelOracle(L * d, d, M, base="rbf", lengthscales=None)
        kern.K = lambda Xt, presliced=False: torch.from_numpy(ko.K(Xt.numpy()))       # stand-in for rank 0's any-shape evaluation
        gram = parallel.ShardedGram(kern, n, torch.device("cpu"), rank, world, chunks=2, ctx=RefusingOnOneRank(rank == refusing_rank))
        out = gram(torch.from_numpy(X))
        ret[rank] = (gram.fallback is not None, None if out is None else float(np.abs(out.numpy() - ko.K(X)).max()))
    finally:
        dist.destroy_process_group()


def test_sharded_gram_ranks_agree_on_the_route_when_one_of_them_is_refused():
    """The ranks all-reduce the verdict of their first chunk: when any single rank's call is refused -- the one that owns no rows
    (n = 20 on 3 ranks of blocks of 8: rank 2 has 4 rows, so refuse rank 1; n = 8: rank 1 and 2 own nothing) or one that does --
    ALL of them take the rank-0 fallback, and the job finishes instead of hanging."""
    for n, world, refusing in ((20, 3, 1), (8, 3, 2), (20, 2, 0)):
        ret = _spawn_with_retry(_diverging_worker, world, (n, refusing))
        assert ret[0] == (True, 0.0), ret
        assert all(ret[r] == (True, None) for r in range(1, world)), ret


class FailingOnOneRank(EmulatorContext):
    """ONE rank's first row-block call fails for good -- a full device (MemoryError from the library's scratch allocation), a HIP error --
    instead of being refused (round 4's advisor: that rank used to raise before the verdict's all-reduce and leave its peers in it)."""

    def __init__(self, fail):
        self.fail = fail

    def call(self, name, p, Xp, n, L, r0, r1, outp):
        if self.fail:
            raise MemoryError("libgpsig_hip: out of device memory")
        if r1 > r0:
            EmulatorContext.call(self, name, p, Xp, n, L, r0, r1, outp)


def _failing_worker(rank, world, port, n, failing_rank, ret):
    import torch
    from gpsig_amd import kernels
    from oracle import sigkern_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, d, M = 7, 2, 3
        rng = np.random.default_rng(6)
        X = np.cumsum(0.3 * rng.standard_normal((n, L, d)), axis=1).reshape(n, L * d)
        kern = kernels.SignatureRBF(L * d, d, M, lengthscales=None)
        ko = O.SignatureKernelOracle(L * d, d, M, base="rbf", lengthscales=None)
        kern.K = lambda Xt, presliced=False: torch.from_numpy(ko.K(Xt.numpy()))       # stand-in for rank 0's any-shape evaluation
        gram = parallel.ShardedGram(kern, n, torch.device("cpu"), rank, world, chunks=2, ctx=FailingOnOneRank(rank == failing_rank))
        try:
            out = gram(torch.from_numpy(X))
            ret[rank] = ("returned", gram.fallback is not None, None if out is None else float(np.abs(out.numpy() - ko.K(X)).max()))
        except MemoryError as e:
            ret[rank] = ("raised", str(e))
    finally:
        dist.destroy_process_group()


def test_sharded_gram_rank_that_fails_for_good_does_not_strand_its_peers():
    """A MemoryError (or any other exception) on one rank's first chunk: that rank votes 'no' in the all-reduce, joins the fallback's
    barrier and raises only then; its peers leave through the rank-0 fallback instead of waiting for the collective's timeout."""
    for n, world, failing in ((20, 2, 1), (20, 3, 0)):
        ret = _spawn_with_retry(_failing_worker, world, (n, failing))
        assert ret[failing][0] == "raised" and "out of device memory" in ret[failing][1], ret
        for r in range(world):
            if r == failing:
                continue
            assert ret[r][0] == "returned" and ret[r][1] is True, ret
            assert (ret[r][2] == 0.0) if r == 0 else (ret[r][2] is None), ret


class FailingOnALaterChunk(EmulatorContext):
    """ONE rank's SECOND row-block call fails (round 5's advisor: out of memory when the scratch buffer grows, a HIP error): the first chunk's
    verdict was unanimous, the peers are inside the chunks' gathers."""

    def __init__(self, fail):
        self.fail, self.calls = fail, 0

    def call(self, name, p, Xp, n, L, r0, r1, outp):
        self.calls += 1
        if self.fail and self.calls >= 2:
            raise MemoryError("libgpsig_hip: out of device memory")
        EmulatorContext.call(self, name, p, Xp, n, L, r0, r1, outp)


def _late_failing_worker(rank, world, port, n, failing_rank, ret):
    import torch
    from gpsig_amd import kernels
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, d, M = 7, 2, 3
        rng = np.random.default_rng(7)
        X = np.cumsum(0.3 * rng.standard_normal((n, L, d)), axis=1).reshape(n, L * d)
        kern = kernels.SignatureRBF(L * d, d, M, lengthscales=None)
        gram = parallel.ShardedGram(kern, n, torch.device("cpu"), rank, world, chunks=3, ctx=FailingOnALaterChunk(rank == failing_rank))
        try:
            gram(torch.from_numpy(X))
            ret[rank] = ("returned",)
        except MemoryError as e:
            ret[rank] = ("raised", "MemoryError", str(e))
        except RuntimeError as e:
            ret[rank] = ("raised", "RuntimeError", str(e))
    finally:
        dist.destroy_process_group()


def test_sharded_gram_failure_on_a_later_chunk_reaches_every_rank():
    """The failing rank keeps joining the gathers its peers are in, then all ranks vote once more: the failing one raises its own error, the
    others raise instead of returning a matrix with that rank's rows missing -- and nobody waits for a collective's timeout."""
    for n, world, failing in ((48, 2, 1), (48, 3, 0)):
        ret = _spawn_with_retry(_late_failing_worker, world, (n, failing))
        assert ret[failing][:2] == ("raised", "MemoryError") and "out of device memory" in ret[failing][2], ret
        for r in range(world):
            if r != failing:
                assert ret[r][:2] == ("raised", "RuntimeError") and "later chunks" in ret[r][2], ret


def _covs_worker(rank, world, port, n, increments, ret):
    import torch
    from gpsig_amd import kernels
    from oracle import sigkern_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, d, M, T = 8, 2, 3, 5
        rng = np.random.default_rng(1)
        X = np.cumsum(0.3 * rng.standard_normal((n, L, d)), axis=1).reshape(n, L * d)
        Z = rng.standard_normal((M * (M + 1) // 2, T, 2, d) if increments else (M * (M + 1) // 2, T, d))
        kern = kernels.SignatureRBF(L * d, d, M, lengthscales=1.3)
        ko = O.SignatureKernelOracle(L * d, d, M, base="rbf", lengthscales=1.3)

        def evaluate(Zb, Xb, inc):          # stand-in for the HIP evaluation of one block (no CPU path in the product)
            return tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in ko.K_tens_n_seq_covs(Zb.numpy(), Xb.numpy(), increments=inc))

        covs = parallel.ShardedCovs(kern, n, torch.device("cpu"), rank, world, evaluate=evaluate)
        out = covs(torch.from_numpy(Z), torch.from_numpy(X), increments=increments)
        if rank == 0:
            want = ko.K_tens_n_seq_covs(Z, X, increments=increments)
            ret["err"] = max(float(np.abs(g.numpy() - w).max()) for g, w in zip(out, want))
            ret["shapes"] = [tuple(g.shape) for g in out]
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,world,increments", [(11, 2, False), (8, 2, True), (2, 3, False)])
def test_gloo_sharded_covariances(n, world, increments):
    """parallel.ShardedCovs: Kzx and the Kxx diagonal in contiguous blocks of sequences per rank, gathered on rank 0 (ragged
    last block; more ranks than whole blocks)."""
    ret = _spawn_with_retry(_covs_worker, world, (n, increments))
    assert ret["err"] == 0.0 and ret["shapes"] == [(5, 5), (5, n), (n,)]


# ---- bench.py --gpus N starts its own ranks (round 3) ------------------------------------------------------------------
def _run_bench(args, env_extra, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    pr = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    return pr.returncode, (json.loads(lines[-1]) if lines else None), pr.stderr


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun: two ranks come up (gloo here), rank 0 reports both of them."""
    rc, line, err = _run_bench(["--gpus", "2", "--rendezvous-check"], {"GPSIG_BENCH_BACKEND": "gloo"})
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 2 and line["ranks_seen"]["world_size"] == 2 and line["ranks_seen"]["backend"] == "gloo"
    assert sorted(r["rank"] for r in line["ranks_seen"]["ranks"]) == [0, 1]
    assert len({r["pid"] for r in line["ranks_seen"]["ranks"]}) == 2
    assert line["ranks_seen"]["launched_by"] == "bench.py itself"


def test_bench_refuses_fewer_gpus_than_ranks():
    """Under the nccl backend a node with fewer GPUs than --gpus must fail, not report a smaller run as n_gpus = N."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        pytest.skip("a node with 64 GPUs")
    rc, line, err = _run_bench(["--gpus", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {"GPSIG_BENCH_BACKEND": "nccl"})
    assert rc != 0 and line is None
    assert "--gpus 64" in err


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    rc, line, err = _run_bench(["--gpus", "1", "--steps", "1", "--no-cpu-baseline"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and line is None and "WORLD_SIZE=2" in err


def test_bench_rank_launch_command():
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cmd = mod.rank_launch_command(8, ["--gpus", "8", "--steps", "5"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "8", "--steps", "5"] and cmd[-5].endswith("bench.py")
