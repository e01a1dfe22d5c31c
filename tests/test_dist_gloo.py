"""world_size-2 gloo test of the multi-GPU protocol (infinite_texture_gans_amd/dist.py) on CPU:
two ranks, each holding half of the images, exchanging exactly what the GPU path exchanges
(fp64 BatchNorm (sum, sumsq) / (sum dy, sum dy*xhat) pairs and one flat gradient all-reduce) must
reproduce the single-process result on the whole batch.  Compute here is torch-CPU (the oracle's
primitives); the product's collectives module is what is under test."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from infinite_texture_gans_amd.dist import SyncGroup, average_flat_gradient, rank_seed, max_over_ranks


class SyncBN(torch.autograd.Function):
    """The exchange pattern of ops._BNAct (stats -> all-reduce -> finalize; bwd sums -> all-reduce),
    restated with torch-CPU ops so that it runs over gloo."""

    @staticmethod
    def forward(ctx, x, gamma, beta, sync):
        n = x.numel() // x.shape[1]
        xd = x.double()
        sums = torch.cat([xd.sum((0, 2, 3)), (xd * xd).sum((0, 2, 3))])
        sync.all_reduce(sums)
        count = sync.global_count(n)
        c = x.shape[1]
        mean = sums[:c] / count
        var = sums[c:] / count - mean * mean
        rstd = (1.0 / torch.sqrt(var + 1e-5)).float()
        mean = mean.float()
        xhat = (x - mean[None, :, None, None]) * rstd[None, :, None, None]
        ctx.save_for_backward(xhat, gamma, rstd)
        ctx.sync, ctx.count = sync, count
        return xhat * gamma[None, :, None, None] + beta[None, :, None, None]

    @staticmethod
    def backward(ctx, dy):
        xhat, gamma, rstd = ctx.saved_tensors
        c = dy.shape[1]
        local = torch.cat([dy.double().sum((0, 2, 3)), (dy.double() * xhat.double()).sum((0, 2, 3))])
        glob = ctx.sync.all_reduce(local.clone())
        m1 = (glob[:c] / ctx.count).float()[None, :, None, None]
        m2 = (glob[c:] / ctx.count).float()[None, :, None, None]
        dx = (gamma * rstd)[None, :, None, None] * (dy - m1 - xhat * m2)
        return dx, local[c:].float(), local[:c].float(), None      # affine grads stay LOCAL sums


def net(x, w1, gamma, beta, w2, bn):
    h = F.conv2d(x, w1, padding=1)
    h = F.leaky_relu(bn(h, gamma, beta), 0.2)
    return F.conv2d(h, w2, padding=1)


def params(seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(6, 3, 3, 3, generator=g) * 0.3, 1 + 0.1 * torch.randn(6, generator=g),
            0.1 * torch.randn(6, generator=g), torch.randn(1, 6, 3, 3, generator=g) * 0.3]


def worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        sync = SyncGroup(dist.group.WORLD)
        assert sync.world == world and sync.global_count(10) == 10 * world
        x_all = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(5))
        x = x_all[rank * 2:(rank + 1) * 2]                 # this rank's images
        ps = [p.clone().requires_grad_(True) for p in params()]
        loss = F.binary_cross_entropy_with_logits(
            net(x, *ps, bn=lambda h, g, b: SyncBN.apply(h, g, b, sync)), torch.full((2, 1, 8, 8), 0.9))
        grads = torch.autograd.grad(loss, ps)
        flat = torch.cat([g.reshape(-1) for g in grads])
        average_flat_gradient(flat, sync)
        t = max_over_ranks(0.5 + rank, torch.device("cpu"), sync)
        assert abs(t - (0.5 + world - 1)) < 1e-12
        assert rank_seed(7, 0) != rank_seed(7, 1)
        if rank == 0:
            torch.save({"flat": flat, "loss": loss.detach()}, out)
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sync_bn_and_flat_grad_allreduce_match_single_process(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(worker, args=(2, free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    # single process, whole batch, plain BatchNorm
    x_all = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(5))
    ps = [p.clone().requires_grad_(True) for p in params()]
    bn = lambda h, g, b: F.batch_norm(h, None, None, g, b, True, 0.1, 1e-5)  # noqa: E731
    loss = F.binary_cross_entropy_with_logits(net(x_all, *ps, bn=bn), torch.full((4, 1, 8, 8), 0.9))
    flat = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, ps)])
    err = float((got["flat"] - flat).norm() / flat.norm())
    assert err < 1e-5, err


def halo_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from infinite_texture_gans_amd.dist import RowHalo
        h = RowHalo(rank, world, dist.group.WORLD)
        assert h.band(7) == [(0, 3), (3, 5), (5, 7)][rank]
        first = torch.full((2, 5, 4), 10.0 * rank + 1)
        last = torch.full((2, 5, 4), 10.0 * rank + 2)
        top, bottom = h.exchange(first, last)
        ok = (top is None) == (rank == 0) and (bottom is None) == (rank == world - 1)
        if top is not None:
            ok = ok and bool((top == 10.0 * (rank - 1) + 2).all())          # the LAST row of the band above
        if bottom is not None:
            ok = ok and bool((bottom == 10.0 * (rank + 1) + 1).all())       # the FIRST row of the band below
        torch.save(ok, out + str(rank))
    finally:
        dist.destroy_process_group()


def test_row_halo_exchange_over_gloo(tmp_path):
    out = str(tmp_path / "ok")
    mp.spawn(halo_worker, args=(3, free_port(), out), nprocs=3, join=True)
    assert all(torch.load(out + str(r)) for r in range(3))
