"""Multi-GPU evaluation of the symmetric Gram K(X): one process per GPU, independent pair blocks,
one RCCL gather.

The reference is single-device; this is the MI355X-native counterpart for
BASELINE.json configs[3].  Every unordered pair {i, j} is owned by exactly one ROW of the Gram (row j
owns the columns i with (j - i) mod N <= N/2), so contiguous row blocks carry equal work, need no
exchange while computing, and the only communication is the gather of the row blocks to rank 0
(each peer sends its block over its own xGMI link), followed by one symmetrisation pass there.
"""
import ctypes as C

from . import _lib

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def row_partition(n, world, align=4):
    """Contiguous row blocks, starts aligned to `align` (the kernel's y-block size)."""
    per = block_rows(n, world, align)
    bounds = [min(r * per, n) for r in range(world + 1)]
    bounds[-1] = n
    return bounds


def block_rows(n, world, align=4):
    """Rows per rank block (the same on every rank, so that one gather of equal-sized tensors moves them)."""
    per = -(-n // world)
    return max(align, -(-per // align) * align)


class ShardedGram:
    """kern.K(X) for X replicated on every rank; the result lands on rank 0 (None elsewhere)."""

    def __init__(self, kern, n, device, rank=0, world=1):
        self.kern, self.n, self.dev, self.rank, self.world = kern, int(n), device, rank, world
        stream = torch.cuda.current_stream(device).cuda_stream
        self.ctx = _lib.context(device.index or 0, stream)
        self.bounds = row_partition(self.n, world)
        if world > 1:
            per = block_rows(self.n, world)
            self.rows = torch.zeros((per, self.n), dtype=torch.float64, device=device)      # equal-sized blocks
            if rank == 0:
                self.half = torch.zeros((per * world, self.n), dtype=torch.float64, device=device)
                self.parts = list(self.half.split(per, dim=0))                                  # gather straight into place
                self.out = torch.empty((self.n, self.n), dtype=torch.float64, device=device)

    def __call__(self, X):
        if self.world == 1:
            return self.kern.K(X)
        keep = []
        p = self.kern._params(keep)
        n, width = X.shape
        L = width // self.kern.num_features
        b0, b1 = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.ctx.set_pointer_mode(_lib.PTR_DEVICE)
        self.ctx.call("gpsig_kernel_K_symm_rows", p, C.c_void_p(X.data_ptr()), n, L, b0, b1, C.c_void_p(self.rows.data_ptr()))
        if dist.get_backend() != "nccl":            # CPU collectives (tests on a box with fewer GPUs than ranks): stage through the host
            torch.cuda.synchronize(self.dev)
            host = self.rows.cpu()
            if self.rank == 0:
                parts = [torch.empty_like(host) for _ in range(self.world)]
                dist.gather(host, gather_list=parts, dst=0)
                for dst_t, src_t in zip(self.parts, parts):
                    dst_t.copy_(src_t)
            else:
                dist.gather(host, dst=0)
                return None
        elif self.rank == 0:
            dist.gather(self.rows, gather_list=self.parts, dst=0)
        else:
            dist.gather(self.rows, dst=0)
            return None
        if self.rank == 0:
            self.ctx.check(self.ctx._lib.gpsig_symmetrize_owned_rows(self.ctx._h, _lib.F64, C.c_void_p(self.half.data_ptr()), n,
                                                                       C.c_void_p(self.out.data_ptr())))
            return self.out
        return None


def symmetrize_reference(half):
    """NumPy statement of gpsig_symmetrize_owned_rows (used by the CPU tests of the partition logic)."""
    import numpy as np
    n = half.shape[0]
    r, c = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    dlt = (r - c) % n
    h = n // 2
    owned = (dlt < h) | ((dlt == h) & ((n % 2 == 1) | (c < r)))
    return np.where(owned, half, half.T), owned
