"""The N > 1 path on CPU: two processes over the gloo backend run the owned-row-block decomposition of the
symmetric Gram (gpsig_amd.parallel) with the kernel replaced by its lock-step CPU emulator, gather the blocks to
rank 0 and symmetrise -- the same partition, ownership rule, task lists and epilogue the 8-GPU run uses."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from gpsig_amd import parallel


def test_row_partition_is_aligned_and_complete():
    for n in (1, 4, 7, 64, 100, 4096, 11584):
        for world in (1, 2, 3, 8):
            b = parallel.row_partition(n, world)
            assert b[0] == 0 and b[-1] == n and len(b) == world + 1
            assert all(b[r] <= b[r + 1] for r in range(world))
            assert all(b[r] % 4 == 0 or b[r] == b[r + 1] for r in range(world))   # aligned, or an empty block


def test_ownership_rule_covers_each_unordered_pair_once():
    for n in (1, 2, 5, 8, 9):
        _, owned = parallel.symmetrize_reference(np.zeros((n, n)))
        off = ~np.eye(n, dtype=bool)
        assert (owned ^ owned.T)[off].all() and owned.diagonal().all()


def _worker(rank, world, port, n, ret):
    import torch
    import emu_util as E
    from oracle import sigkern_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, d, M = 9, 3, 3
        rng = np.random.default_rng(0)
        X = np.cumsum(0.3 * rng.standard_normal((n, L, d)), axis=1)       # replicated input
        var = np.array([0.7, 1.1, 0.9, 1.3])
        b = parallel.row_partition(n, world)
        per = parallel.block_rows(n, world)
        t = torch.zeros((per, n), dtype=torch.float64)                       # equal-sized blocks, as ShardedGram allocates
        t[: b[rank + 1] - b[rank]] = torch.from_numpy(E.kernel_K_owned_rows(X, "rbf", M, var, 1.0, True, b[rank], b[rank + 1]))
        if rank == 0:
            half_t = torch.zeros((per * world, n), dtype=torch.float64)
            dist.gather(t, gather_list=list(half_t.split(per, dim=0)), dst=0)
            half = half_t.numpy()[:n]
            full, _ = parallel.symmetrize_reference(half)
            want = O.SignatureKernelOracle(L * d, d, M, base="rbf", variances=var, lengthscales=None).K(X.reshape(n, -1))
            ret["err"] = float(np.abs(full - want).max() / np.abs(want).max())
        else:
            dist.gather(t, dst=0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [22, 37])
def test_two_rank_gloo_owned_rows_gather(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, n, ret), nprocs=2, join=True)
        assert ret["err"] < 1e-12
