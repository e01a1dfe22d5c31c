"""Multi-GPU protocol of the train step: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm; the same code runs over gloo on CPU tensors, which is how the protocol is
tested without GPUs).

The reference's multi-GPU path is single-process nn.DataParallel with per-replica BatchNorm
statistics (reference train.py:74-77).  This build shards the work units (images, and with them
their 3x3 patch grids) across ranks:

  * BatchNorm: per-rank statistics by default (the reference's replica semantics).  With sync-BN
    (engine.Trainer(sync_bn=True) / --sync_bn) every rank reduces its pixels to per-channel fp64
    (sum, sumsq) pairs and ONE all-reduce of 2*C doubles per norm layer makes the statistics global -
    the SINGLE-PROCESS semantics of a batch that is world-size times larger.  Backward likewise
    all-reduces (sum dy, sum dy*xhat); the affine-parameter gradients stay local sums: they are added up
    across ranks by the gradient all-reduce like every other parameter gradient.
  * gradients: each model's parameters live in one flat fp32 buffer, so the exchange is ONE
    all-reduce per model per step (G 21 MB, D 11 MB), followed by 1/world scaling because each rank's
    loss is a mean over its own shard.
  * no other data-path collective exists: latents and real crops are generated per rank from
    rank-dependent seeds.
"""
import torch


def _active(sync):
    """Collectives are issued when there is more than one rank - or, for rehearsing the RCCL path on a single GPU,
    when ITG_FORCE_COLLECTIVES=1 (a one-rank all-reduce leaves the data unchanged)."""
    return sync.world > 1 or _FORCE


import os as _os
_FORCE = _os.environ.get("ITG_FORCE_COLLECTIVES", "0") == "1"
_HALO_STREAM = _os.environ.get("ITG_HALO_STREAM", "0") == "1"      # halo rows on a communication stream (RowHalo.exchange)


class SyncGroup:
    """Process group over which per-channel statistics are summed."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)

    def all_reduce(self, t):
        """In-place SUM over ranks (fp64 statistics buffers)."""
        if self.world > 1 or _FORCE:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def global_count(self, local_count):
        """Pixel count behind the all-reduced statistics (every rank holds the same shard size)."""
        return local_count * self.world


def average_flat_gradient(flat_grad, sync):
    """ONE all-reduce of a model's flat gradient buffer, then the 1/world factor of the global mean."""
    if sync is not None and _active(sync):
        sync.all_reduce(flat_grad)
        flat_grad.mul_(1.0 / sync.world)
    return flat_grad


def rank_seed(base_seed, rank, stream=0):
    """Disjoint RNG streams per rank for synthetic inputs / latents."""
    return int(base_seed) + 1 + int(rank) + 1000 * int(stream)


def max_over_ranks(seconds, device, sync):
    """Wall time of the slowest rank (the bench contract's max-over-ranks timing)."""
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    if sync is not None and _active(sync):
        sync.dist.all_reduce(t, op=sync.dist.ReduceOp.MAX, group=sync.group)
    return float(t)


# ------------------------------------------------------------------------------- patch-grid row sharding
class RowHalo:
    """Halo exchange for a patch grid sharded by patch ROWS over ranks (rank r owns a contiguous band
    of patch rows; outer borders stay local).  Before every 3x3 conv each rank sends its first pixel
    row up and its last pixel row down and receives the neighbours' rows: two messages of
    n * W * ld floats per conv layer and direction, posted together (batch_isend_irecv -> one RCCL
    group call over the two xGMI links involved).  Rows are (n, W, ld) fp32 tensors."""

    def __init__(self, rank, world, group=None):
        self.rank, self.world, self.group = rank, world, group

    def band(self, total_rows):
        """[a, b) patch rows of this rank (contiguous, as equal as possible, none empty)."""
        if self.world > total_rows:
            raise ValueError("cannot shard %d patch rows over %d ranks" % (total_rows, self.world))
        base, extra = divmod(total_rows, self.world)
        a = self.rank * base + min(self.rank, extra)
        return a, a + base + (1 if self.rank < extra else 0)

    _pending = None        # (consumer stream, communication stream) of an exchange posted with defer_wait

    def wait(self):
        """The consumer stream waits for the exchange posted with ``defer_wait`` (no-op otherwise)."""
        if self._pending is not None:
            cur, comm = self._pending
            self._pending = None
            cur.wait_stream(comm)

    def exchange(self, first_row, last_row, defer_wait=False):
        """-> (top, bottom): the row above this band / below it, or None at the grid's outer border.
        ``defer_wait`` (with the communication stream, ITG_HALO_STREAM=1): the transfers are posted but the consumer stream
        does not wait yet - the caller launches what needs no halo row and then calls wait().  Without a communication
        stream (default, gloo, host staging) the exchange is complete on return and wait() is a no-op."""
        import torch.distributed as dist
        if self.world == 1:
            return None, None
        if first_row.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal of the protocol with several ranks on ONE GPU (gloo has no device send/recv):
            # stage the rows through host memory.  RCCL runs take the direct path below.
            top, bottom = self.exchange(first_row.cpu(), last_row.cpu())
            dev = first_row.device
            return (None if top is None else top.to(dev)), (None if bottom is None else bottom.to(dev))
        ops, top, bottom = [], None, None
        if self.rank > 0:
            top = torch.empty_like(first_row)
            ops += [dist.P2POp(dist.isend, first_row, self.rank - 1, self.group),
                    dist.P2POp(dist.irecv, top, self.rank - 1, self.group)]
        if self.rank < self.world - 1:
            bottom = torch.empty_like(last_row)
            ops += [dist.P2POp(dist.isend, last_row, self.rank + 1, self.group),
                    dist.P2POp(dist.irecv, bottom, self.rank + 1, self.group)]
        if not ops:
            return top, bottom
        if first_row.is_cuda and _HALO_STREAM:
            # Device rows (RCCL), OPT-IN (ITG_HALO_STREAM=1) until it has run on a real multi-rank RCCL group once (ADVICE r3:
            # no gloo test can take this branch): the four transfers are posted as one group on a communication stream of
            # this object and the CONSUMER stream waits for that stream - the host never blocks (ProcessGroupNCCL's
            # Work.wait() is a stream-level wait as well, but it is taken on whatever stream is current; an explicit
            # communication stream keeps the transfers off the compute queue and lets whatever the caller issues before it
            # needs the rows run beside them).  The rows are kept referenced until the consumer has waited.
            # Default: the plain form below on the current stream - the same calls the gloo tests execute.
            cur = torch.cuda.current_stream(first_row.device)
            comm = self._comm_stream(first_row.device)
            comm.wait_stream(cur)                          # the rows to send are produced on the compute stream
            with torch.cuda.stream(comm):
                for req in dist.batch_isend_irecv(ops):
                    req.wait()                             # stream-level: orders `comm` behind the transfer
            if defer_wait:
                self._pending = (cur, comm)
            else:
                cur.wait_stream(comm)
            for t in (first_row, last_row, top, bottom):
                if t is not None:
                    t.record_stream(comm)
        else:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return top, bottom

    _comm = None

    def _comm_stream(self, device):
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=device)
        return self._comm


class ThreadRowHalo(RowHalo):
    """Same protocol between Python threads of ONE process (all bands on one GPU): used to validate the
    sharded path against the unsharded result on a single-GPU box."""

    class Shared:
        def __init__(self, world):
            import threading
            self.world = world
            self.barrier = threading.Barrier(world)
            self.first = [None] * world
            self.last = [None] * world

    def __init__(self, rank, shared):
        super().__init__(rank, shared.world)
        self.shared = shared

    def exchange(self, first_row, last_row, defer_wait=False):
        s = self.shared
        s.first[self.rank], s.last[self.rank] = first_row, last_row
        torch.cuda.synchronize() if first_row.is_cuda else None
        s.barrier.wait()
        top = s.last[self.rank - 1].clone() if self.rank > 0 else None
        bottom = s.first[self.rank + 1].clone() if self.rank < self.world - 1 else None
        s.barrier.wait()
        return top, bottom


# ------------------------------------------------------------------------------- row-sharded TRAINING
class BandComm(RowHalo):
    """Collectives of a train step whose patch grid is sharded by rows (BASELINE config 4):
    halo rows per conv (exchange, also used by the backward for the halo gradients), sync-BN sums and
    flat-gradient averaging (all_reduce), and the band -> full-image gather in front of the
    discriminator (all_gather).  Backed by torch.distributed (RCCL on GPUs, gloo on CPU)."""

    def __init__(self, rank, world, group=None):
        super().__init__(rank, world, group)
        import torch.distributed as dist
        self.dist = dist

    def all_reduce(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def global_count(self, local_count):
        """Images are split evenly over ranks in front of the discriminator."""
        return local_count * self.world

    def band_sync(self, total_rows):
        """Statistics group for layers that run on this rank's band of ``total_rows`` patch rows."""
        a, b = self.band(total_rows)
        return _BandSync(self, total_rows, b - a)

    def all_gather(self, t, heights=None):
        """list of every rank's tensor.  ``heights``: the size of dim -2 on every rank when the caller knows it (the band
        heights of a row-sharded grid are static: band() x patch height) - otherwise the sizes are exchanged first, which
        costs a host synchronisation per call."""
        if self.world == 1:
            return [t]
        if t.is_cuda and self.dist.get_backend(self.group) == "gloo":      # one-GPU rehearsal, see exchange()
            return [o.to(t.device) for o in self.all_gather(t.cpu(), heights)]
        if heights is None:
            sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(self.world)]
            self.dist.all_gather(sizes, torch.tensor([t.shape[-2]], dtype=torch.int64, device=t.device), group=self.group)
            heights = [int(s) for s in sizes]
        bad = heights[self.rank] != t.shape[-2]
        if bad:
            # a rank that raised here would leave the others blocked inside the collective: take part with a
            # correctly shaped stand-in, then fail - the launcher tears the job down from this rank's error
            mine, t = t.shape[-2], t.new_zeros(t.shape[:-2] + (heights[self.rank], t.shape[-1]))
        outs = [torch.empty(t.shape[:-2] + (h, t.shape[-1]), dtype=t.dtype, device=t.device) for h in heights]
        if all(h == t.shape[-2] for h in heights):
            self.dist.all_gather(outs, t.contiguous(), group=self.group)
        else:
            for r in range(self.world):      # ragged bands: one broadcast per rank
                if r == self.rank:
                    outs[r].copy_(t)
                self.dist.broadcast(outs[r], src=r, group=self.group)
        if bad:
            raise ValueError("rank %d: band of %d rows announced as %d" % (self.rank, mine, heights[self.rank]))
        return outs

    def reduce_scatter_rows(self, g, heights):
        """SUM over ranks of ``g`` (..., H, W), returning only this rank's band of rows: with equal bands one
        reduce-scatter (each rank receives 1/world of the tensor instead of all of it), with ragged bands the all-reduce
        followed by the slice."""
        lo = sum(heights[:self.rank])
        h = heights[self.rank]
        if self.world == 1:
            return g[..., lo:lo + h, :].contiguous()
        if g.is_cuda and self.dist.get_backend(self.group) == "gloo":
            return self.reduce_scatter_rows(g.cpu(), heights).to(g.device)
        if len(set(heights)) == 1:          # (gloo implements reduce_scatter too: the CPU tests take this very branch)
            chunks, o = [], 0
            for hr in heights:
                chunks.append(g[..., o:o + hr, :].contiguous())
                o += hr
            out = torch.empty_like(chunks[self.rank])
            self.dist.reduce_scatter(out, chunks, op=self.dist.ReduceOp.SUM, group=self.group)
            return out
        g = g.contiguous().clone()
        self.dist.all_reduce(g, op=self.dist.ReduceOp.SUM, group=self.group)
        return g[..., lo:lo + h, :].contiguous()

    def band_heights(self, total_rows, patch):
        """Pixel-row count of every rank's band: static for a model (no exchange needed)."""
        out = []
        for r in range(self.world):
            base, extra = divmod(total_rows, self.world)
            out.append((base + (1 if r < extra else 0)) * patch)
        return out


class _BandSync:
    """BatchNorm statistics over all bands: sums are all-reduced, and the pixel count scales by
    total_rows / my_rows (bands may differ in height, every pixel row is equally wide)."""

    def __init__(self, comm, total_rows, my_rows):
        self.comm, self.world, self.total, self.mine = comm, comm.world, total_rows, my_rows

    def all_reduce(self, t):
        return self.comm.all_reduce(t)

    def global_count(self, local_count):
        assert (local_count * self.total) % self.mine == 0
        return local_count * self.total // self.mine
