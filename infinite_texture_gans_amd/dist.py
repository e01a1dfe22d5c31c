"""Multi-GPU protocol of the train step: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm; the same code runs over gloo on CPU tensors, which is how the protocol is
tested without GPUs).

The reference's multi-GPU path is single-process nn.DataParallel with per-replica BatchNorm
statistics (reference train.py:74-77).  This build shards the work units (images, and with them
their 3x3 patch grids) across ranks and keeps the SINGLE-PROCESS semantics of a batch that is
world-size times larger:

  * BatchNorm: every rank reduces its pixels to per-channel fp64 (sum, sumsq) pairs; ONE all-reduce of
    2*C doubles per norm layer makes the statistics global.  Backward likewise all-reduces
    (sum dy, sum dy*xhat).  The affine-parameter gradients stay local sums: they are added up across
    ranks by the gradient all-reduce like every other parameter gradient.
  * gradients: each model's parameters live in one flat fp32 buffer, so the exchange is ONE
    all-reduce per model per step (G 21 MB, D 11 MB), followed by 1/world scaling because each rank's
    loss is a mean over its own shard.
  * no other data-path collective exists: latents and real crops are generated per rank from
    rank-dependent seeds.
"""
import torch


class SyncGroup:
    """Process group over which per-channel statistics are summed."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)

    def all_reduce(self, t):
        """In-place SUM over ranks (fp64 statistics buffers)."""
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def global_count(self, local_count):
        """Pixel count behind the all-reduced statistics (every rank holds the same shard size)."""
        return local_count * self.world


def average_flat_gradient(flat_grad, sync):
    """ONE all-reduce of a model's flat gradient buffer, then the 1/world factor of the global mean."""
    if sync is not None and sync.world > 1:
        sync.dist.all_reduce(flat_grad, op=sync.dist.ReduceOp.SUM, group=sync.group)
        flat_grad.mul_(1.0 / sync.world)
    return flat_grad


def rank_seed(base_seed, rank, stream=0):
    """Disjoint RNG streams per rank for synthetic inputs / latents."""
    return int(base_seed) + 1 + int(rank) + 1000 * int(stream)


def max_over_ranks(seconds, device, sync):
    """Wall time of the slowest rank (the bench contract's max-over-ranks timing)."""
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    if sync is not None and sync.world > 1:
        sync.dist.all_reduce(t, op=sync.dist.ReduceOp.MAX, group=sync.group)
    return float(t)
