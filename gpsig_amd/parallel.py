"""Multi-GPU evaluation of the symmetric Gram K(X): one process per GPU, independent pair blocks,
one gather (RCCL over xGMI) overlapped with the computation.

The reference is single-device; this is the MI355X-native counterpart for
BASELINE.json configs[3] (N = 32768 over 8 GPUs).  Every unordered pair {i, j} is owned by exactly one
ROW of the Gram -- row j owns the N/2+1 columns j-N/2 .. j (mod N) -- so contiguous row blocks carry
equal work and need no exchange while computing.  A rank writes the owned entries of its rows side by
side (`gpsig_kernel_K_symm_rows_compact`: a (rows, N/2+1) block, half the bytes of full rows), in a few
chunks; each finished chunk is handed to an asynchronous `dist.gather` (every peer sends over its own
xGMI link to rank 0) while the next chunk is being computed, and rank 0 turns the stacked blocks into
the full symmetric matrix in one tiled pass (`gpsig_symmetrize_compact_rows`).
"""
import ctypes as C
import time

from . import _lib

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None

ALIGN = 4   # the pair kernel's y-block size (64 lanes / 16 lanes per pair): row ranges start at multiples of it


def block_rows(n, world, align=ALIGN):
    """Rows per rank block (the same on every rank, so that gathers move equal-sized tensors)."""
    per = -(-n // world)
    return max(align, -(-per // align) * align)


def row_partition(n, world, align=ALIGN):
    """Contiguous row blocks, starts aligned to `align`."""
    per = block_rows(n, world, align)
    bounds = [min(r * per, n) for r in range(world + 1)]
    bounds[-1] = n
    return bounds


class _Stopwatch:
    """Marks on the current stream (CUDA events) or on the host clock (CPU tensors under gloo): where a sharded Gram's time went on THIS
    rank.  Recording an event does not synchronise anything; elapsed() does, once, when somebody asks."""

    def __init__(self, dev):
        self.cuda = dev.type == "cuda"
        self.dev = dev
        self.marks = {}

    def mark(self, name):
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.dev))
            self.marks[name] = ev
        else:
            self.marks[name] = time.perf_counter()

    def elapsed(self, a, b):
        if a not in self.marks or b not in self.marks:
            return 0.0
        if self.cuda:
            self.marks[b].synchronize()
            return float(self.marks[a].elapsed_time(self.marks[b]))
        return (self.marks[b] - self.marks[a]) * 1e3


class ShardedGram:
    """kern.K(X) for X replicated on every rank; the result lands on rank 0 (None elsewhere).

    chunks: pieces a rank's row block is computed and gathered in (the gather of piece k runs while piece k+1 is computed).
    ctx:    the library context to use (default: the one of `device` and the stream current at call time).  The CPU test-suite
            passes a stand-in that runs the kernel's lock-step emulator, so that this very code runs under gloo without a GPU.
    Shapes the row-block kernels are not built for (gpsig_amd/csrc/seq_configs.hpp: both sequences longer than the register-resident
    side holds, wide state spaces, ...) are evaluated by rank 0 alone through kern.K and its any-shape kernels (`self.fallback` then
    says why); every rank takes that branch together, the shape being the same everywhere.

    Stream ordering the RCCL branch relies on (torch.distributed's ProcessGroupNCCL): the library's kernels run on the stream that is
    current when __call__ is entered (the context is looked up by it); `dist.gather(..., async_op=True)` makes the collective's own
    stream wait for an event recorded on that current stream at the time of the call -- so the kernel that wrote chunk k is ordered
    before the gather that sends it, while chunk k+1, launched afterwards on the current stream, overlaps with it (it writes other rows
    of `self.rows`); `work.wait()` makes the CURRENT STREAM (not the host) wait for the collective, after which rank 0's
    symmetrisation, launched on the current stream, reads `self.half`.  `self.rows` / `self.half` / `self.out` live as long as the
    object, so nothing is handed back to the caching allocator while a collective may still read it; a second __call__ overwrites
    `self.rows` only after the waits of the first."""

    def __init__(self, kern, n, device, rank=0, world=1, chunks=4, ctx=None, force=False):
        self.kern, self.n, self.dev, self.rank, self.world = kern, int(n), torch.device(device), int(rank), int(world)
        self._ctx = ctx
        # force: take the decomposed route (row-block calls, the verdict all-reduce, asynchronous gathers, symmetrisation) on ONE rank
        # too -- a single MI355X then runs every collective of the N-rank path through RCCL itself (tests/test_gpu_parity.py)
        self.force = bool(force)
        self.fallback = None                     # why rank 0 evaluated alone, when it did
        self._watch = None                       # marks of the last call (timings())
        self.width = self.n // 2 + 1
        chunks = max(1, int(chunks))
        self.per = block_rows(self.n, world, ALIGN * chunks)
        self.chunk_rows = self.per // chunks
        self.chunks = chunks
        self.bounds = row_partition(self.n, world, ALIGN * chunks)
        if world > 1 or self.force:
            self.rows = torch.zeros((self.per, self.width), dtype=torch.float64, device=self.dev)        # equal-sized blocks
            if rank == 0:
                self.half = torch.zeros((self.per * world, self.width), dtype=torch.float64, device=self.dev)
                self.out = torch.empty((self.n, self.n), dtype=torch.float64, device=self.dev)

    def _context(self):
        if self._ctx is not None:
            return self._ctx
        ctx = _lib.context(self.dev.index or 0, torch.cuda.current_stream(self.dev).cuda_stream)
        ctx.set_pointer_mode(_lib.PTR_DEVICE)
        return ctx

    def _gather(self, k):
        """Start the gather of chunk k of every rank's block into rank 0's `half`; returns the pending work (or None)."""
        cr = self.chunk_rows
        mine = self.rows[k * cr:(k + 1) * cr]
        parts = None
        if self.rank == 0:
            parts = [self.half[r * self.per + k * cr: r * self.per + (k + 1) * cr] for r in range(self.world)]
        # what a gather moves must be one contiguous block of the same size on every rank (block_rows() makes it so: `per` and
        # `chunk_rows` depend on (n, world, chunks) only); a strided or short operand would be copied or mis-sized silently
        assert mine.is_contiguous() and tuple(mine.shape) == (cr, self.width), (tuple(mine.shape), mine.is_contiguous())
        assert parts is None or all(p_.is_contiguous() and p_.shape == mine.shape for p_ in parts)
        if mine.is_cuda and dist.get_backend() != "nccl":
            # CPU collectives on GPU data (tests on a box with fewer GPUs than ranks): stage through the host
            torch.cuda.synchronize(self.dev)
            host = mine.cpu()
            if self.rank == 0:
                hp = [torch.empty_like(host) for _ in range(self.world)]
                dist.gather(host, gather_list=hp, dst=0)
                for dst_t, src_t in zip(parts, hp):
                    dst_t.copy_(src_t)
            else:
                dist.gather(host, dst=0)
            return None
        return dist.gather(mine, gather_list=parts, dst=0, async_op=True)

    def __call__(self, X):
        if self.world == 1 and not self.force:
            return self.kern.K(X)
        X, _ = self.kern._slice(X, None)
        if X.dtype != torch.float64 or X.device != self.dev:
            raise ValueError("ShardedGram takes a float64 tensor on %s" % (self.dev,))
        X = X.contiguous()
        n, L = self.kern._seq_dims(X)
        if n != self.n:
            raise ValueError("ShardedGram was built for %d sequences, got %d" % (self.n, n))
        keep = []
        p = self.kern._params(keep)
        ctx = self._context()
        b0, b1 = self.bounds[self.rank], self.bounds[self.rank + 1]
        pending = []
        # the chunks of this block are row-block calls on the same X: SignatureLinear's feature matrix (csrc/sig_feat_kernel.hpp) is
        # built by the first and kept for the others (7.6 ms per call at N = 32,768 against 18 ms of contraction per chunk on 8 ranks)
        keep_features = getattr(ctx, "set_option", None)
        if keep_features is not None:
            keep_features("sig_features_keep", 1)
        try:
            return self._chunks(ctx, p, X, n, L, b0, b1, pending)
        finally:
            if keep_features is not None:
                keep_features("sig_features_keep", 0)

    def _agree(self, ok):
        """True where EVERY rank's first chunk was taken by the row-block kernels.  One scalar all-reduce (MIN) per Gram: a rank whose
        call was refused (or, on a full device, took another decision than its peers) must not leave the others inside a gather that
        it never joins -- whatever the library decides per rank, the ranks leave this function on the same branch."""
        nccl = dist.get_backend() == "nccl"
        v = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.dev if nccl else "cpu")
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        return bool(int(v.item()))

    def _chunks(self, ctx, p, X, n, L, b0, b1, pending):
        sw = self._watch = _Stopwatch(self.dev)
        sw.chunks = 0
        late = None                    # this rank's failure on a chunk after the first
        for k in range(self.chunks):
            r0 = min(b0 + k * self.chunk_rows, b1)
            r1 = min(r0 + self.chunk_rows, b1)
            sw.mark("b%d" % k)
            if (r1 > r0 or k == 0) and late is None:
                blk = self.rows[k * self.chunk_rows:]
                why, err = None, None
                try:
                    # (an empty row range on the first chunk still asks the library's routes for this shape, in the order a block with
                    # rows does: api.hip, e_kernel_K_symm_rows)
                    ctx.call("gpsig_kernel_K_symm_rows_compact", p, C.c_void_p(X.data_ptr()), n, L, r0, r1, C.c_void_p(blk.data_ptr()))
                except NotImplementedError as e:
                    if k != 0:
                        late = late or e
                    why = str(e)
                except Exception as e:      # noqa: BLE001 -- a full device (MemoryError), a HIP error: this rank cannot go on, but its peers
                    if k != 0:              # are about to enter the verdict's all-reduce and must not wait there for the collective's timeout
                        late = late or e    # (round 6) a later chunk: keep joining the gathers the peers are in, vote after the last one
                    why, err = repr(e), e
                if k == 0 and not self._agree(why is None):
                    return self._rank0_alone(X, why or "another rank's row-block call was refused", err)
            sw.mark("a%d" % k)
            sw.chunks = k + 1
            w = self._gather(k)       # enqueued behind chunk k on the collective's own stream; chunk k+1 starts meanwhile
            if w is not None:
                pending.append(w)
        sw.mark("e0")
        for w in pending:
            w.wait()
        sw.mark("e1")
        # One more scalar vote where there was more than one chunk: a rank that failed after its first chunk (out of memory when its scratch grew,
        # a HIP error) has still joined every gather -- with whatever its buffer held -- so nobody is left inside a collective; now every rank
        # learns of it and none returns a matrix with that rank's rows missing.  (A sticky HIP error can break the failing rank's own
        # collectives under RCCL: then its peers still depend on the collective's timeout.)
        if self.chunks > 1 and not self._agree(late is None):
            self._watch = None
            raise late if late is not None else RuntimeError("ShardedGram: another rank failed in one of its later chunks")
        if self.rank != 0:
            return None
        ctx.symmetrize_compact_rows(_lib.F64, C.c_void_p(self.half.data_ptr()), n, C.c_void_p(self.out.data_ptr()))
        sw.mark("e2")
        return self.out

    def timings(self):
        """Where the last decomposed call's time went on this rank, in ms of the stream it ran on (host clock for CPU tensors):
        compute_ms -- the row-block kernels of its chunks; gather_inline_ms -- what the stream spent between chunks starting the gathers
        (~0 for RCCL's asynchronous gathers, the whole staged transfer under gloo); gather_wait_ms -- what it waited for the gathers after
        its last chunk; symmetrise_ms -- rank 0's pass from compact rows to the full matrix.  Synchronises with the events it reads.
        None when the last call did not take the decomposed route."""
        sw = self._watch
        if sw is None or "e1" not in sw.marks:
            return None
        nk = sw.chunks
        comp = sum(sw.elapsed("b%d" % k, "a%d" % k) for k in range(nk))
        inline = sum(sw.elapsed("a%d" % k, "b%d" % (k + 1)) for k in range(nk - 1)) + sw.elapsed("a%d" % (nk - 1), "e0")
        return {"rank": self.rank, "rows": int(self.bounds[self.rank + 1] - self.bounds[self.rank]), "chunks": nk, "compute_ms": comp,
                "gather_inline_ms": inline, "gather_wait_ms": sw.elapsed("e0", "e1"),
                "symmetrise_ms": sw.elapsed("e1", "e2") if "e2" in sw.marks else 0.0}

    def _rank0_alone(self, X, why, err=None):
        """A shape the row-block kernels do not take: rank 0 evaluates K(X) through the any-shape kernels, the others wait for it.
        err: this rank's own call failed for good (not a refusal): it still joins the barrier its peers wait in, then raises."""
        self.fallback = why
        self._watch = None
        out = None
        try:
            if err is None and self.rank == 0:
                out = self.kern.K(X, presliced=True)
        finally:
            if dist is not None and dist.is_initialized():
                dist.barrier()
        if err is not None:
            raise err
        return out


class ShardedCovs:
    """The three SVGP covariances of ``kern.K_tens_n_seq_covs(Z, X)`` (gpsig/kernels.py:591-671; BASELINE configs[2]) with the N
    sequences split over the ranks: the (tensor, sequence) chains are independent, so rank r evaluates Kzx[:, n_r] and the
    Kxx diagonal of its contiguous block of sequences (Z is replicated: a few hundred KB) and the blocks are gathered on rank 0
    -- the "N x M batch shards embarrassingly" half of the north star; no exchange while computing.  Kzz (T x T, independent of
    X: 0.02 ms at T = 512) comes out of the same call on every rank and is used on rank 0 only.  Returns (Kzz, Kzx, Kxx_diag) on
    rank 0, None elsewhere.

    evaluate: what computes a block, ``kern.K_tens_n_seq_covs`` by default (the CPU test-suite passes a stand-in: there is no CPU
    path in the product)."""

    def __init__(self, kern, n, device, rank=0, world=1, evaluate=None, force=False):
        self.force = bool(force)        # a one-rank group goes through the collectives too (the RCCL path on a single-GPU test box)
        self.kern, self.n, self.dev, self.rank, self.world = kern, int(n), torch.device(device), int(rank), int(world)
        self.per = -(-self.n // self.world)                    # equal-sized blocks (the last one padded) so that gathers are regular
        self.n0 = min(self.rank * self.per, self.n)
        self.n1 = min(self.n0 + self.per, self.n)
        self._evaluate = evaluate

    def __call__(self, Z, X, increments=False):
        f = self._evaluate or (lambda Zb, Xb, inc: self.kern.K_tens_n_seq_covs(Zb, Xb, increments=inc))
        if self.world == 1 and not self.force:
            return f(Z, X, increments)
        if X.shape[0] != self.n:
            raise ValueError("ShardedCovs was built for %d sequences, got %d" % (self.n, X.shape[0]))
        if torch.is_tensor(X) and self._evaluate is None and (not X.is_cuda or X.device != self.dev or Z.device != self.dev):
            raise ValueError("ShardedCovs takes tensors on %s" % (self.dev,))
        if X.dtype != Z.dtype:
            raise ValueError("ShardedCovs takes Z and X of one dtype (got %s and %s)" % (Z.dtype, X.dtype))
        T = Z.shape[1]
        if self.n1 > self.n0:
            Kzz, Kzx, Kxx = f(Z, X[self.n0:self.n1].contiguous(), increments)
        else:                                                   # more ranks than sequences
            Kzz = Kzx = Kxx = None
        # one regular block per rank: [Kzx block transposed | Kxx-diag block], (per, T + 1), zero beyond the rank's sequences
        blk = torch.zeros((self.per, T + 1), dtype=X.dtype, device=X.device)
        if Kzx is not None:
            blk[: self.n1 - self.n0, :T] = Kzx.T
            blk[: self.n1 - self.n0, T] = Kxx
        parts = [torch.empty_like(blk) for _ in range(self.world)] if self.rank == 0 else None
        if blk.is_cuda and dist.get_backend() != "nccl":        # CPU collectives on GPU data: stage through the host (tests)
            host = blk.cpu()
            hp = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(host, gather_list=hp, dst=0)
            if self.rank == 0:
                for d_, s_ in zip(parts, hp):
                    d_.copy_(s_)
        else:
            dist.gather(blk, gather_list=parts, dst=0)
        if self.rank != 0:
            return None
        allb = torch.cat(parts, dim=0)[: self.n]
        return Kzz, allb[:, :T].T.contiguous(), allb[:, T].contiguous()


def owned_mask(n):
    """owned[r, c]: row r owns column c (the emission predicate PRED_CIRCULANT of csrc/seq_args.hpp with i = c, j = r)."""
    import numpy as np
    r, c = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    dlt = (r - c) % n
    h = n // 2
    return (dlt < h) | ((dlt == h) & ((n % 2 == 1) | (c < r)))


def symmetrize_reference(half):
    """NumPy statement of gpsig_symmetrize_owned_rows (used by the CPU tests of the partition logic)."""
    import numpy as np
    owned = owned_mask(half.shape[0])
    return np.where(owned, half, half.T), owned


def symmetrize_compact_reference(half):
    """NumPy statement of gpsig_symmetrize_compact_rows: half (n, n//2+1) -> (n, n)."""
    import numpy as np
    n = half.shape[0]
    h = n // 2
    owned = owned_mask(n)
    r, c = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    own_rows = half[r, np.clip(h - (r - c) % n, 0, h)]          # valid where owned
    own_cols = half[c, np.clip(h - (c - r) % n, 0, h)]          # valid where the column's row owns the entry
    return np.where(owned, own_rows, own_cols)
