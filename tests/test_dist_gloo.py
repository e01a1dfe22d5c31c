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


def _band_net(x, w1, w2, halo=None):
    """Two replicate-padded 3x3 convs; with ``halo`` the rows above/below come from the neighbours."""
    from infinite_texture_gans_amd import ops
    h = x
    for w in (w1, w2):
        top = bottom = None
        if halo is not None:
            top, bottom = ops.halo_exchange(h[:, :, 0], h[:, :, -1], halo)
        top = h[:, :, 0] if top is None else top
        bottom = h[:, :, -1] if bottom is None else bottom
        ext = torch.cat((top.unsqueeze(2), h, bottom.unsqueeze(2)), 2)
        h = torch.tanh(F.conv2d(F.pad(ext, (1, 1, 0, 0), mode="replicate"), w))
    return h


def band_worker(rank, world, port, out, total_rows=7):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from infinite_texture_gans_amd.dist import BandComm
        from infinite_texture_gans_amd import ops
        torch.set_num_threads(1)
        comm = BandComm(rank, world, dist.group.WORLD)
        g = torch.Generator().manual_seed(11)
        H = 2 * total_rows
        x = torch.randn(world, 2, H, 6, generator=g)
        w1 = (torch.randn(4, 2, 3, 3, generator=g) * 0.4).requires_grad_(True)
        w2 = (torch.randn(1, 4, 3, 3, generator=g) * 0.4).requires_grad_(True)
        a, b = comm.band(total_rows)                          # 7 rows on 3 ranks: ragged, 3 + 2 + 2 patch rows of 2 pixels
        sync = comm.band_sync(total_rows)
        assert sync.global_count(world * (b - a) * 2 * 6) == world * H * 6
        band = _band_net(x[:, :, 2 * a:2 * b], w1, w2, comm)
        # static heights: equal bands take the reduce-scatter branch of the gather's backward, ragged ones all-reduce + slice
        full = ops.gather_rows(band, comm, comm.band_heights(total_rows, 2))      # whole images on every rank
        assert full.shape == (world, 1, H, 6)
        mine = full[rank:rank + 1]                            # "D" scores one image per rank
        loss = (mine * torch.linspace(-1, 1, H * 6).reshape(1, 1, H, 6)).sum() + (mine ** 2).mean()
        flat = torch.cat([t.reshape(-1) for t in torch.autograd.grad(loss, (w1, w2))])
        average_flat_gradient(flat, comm)
        if rank == 0:
            torch.save({"flat": flat, "full": full.detach()}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total_rows", [(3, 7), (2, 8)], ids=["3_ragged_bands", "2_equal_bands_reduce_scatter"])
def test_band_sharded_forward_backward_matches_single_process(tmp_path, world, total_rows):
    """Ranks with ragged (3 + 2 + 2) or equal (4 + 4) bands: differentiable halo exchange (halo gradients travel back),
    band gather with summed band gradients (equal bands: the reduce-scatter branch of dist.BandComm.reduce_scatter_rows, the
    one an RCCL group takes too; ragged: all-reduce + slice), flat gradient averaging == the unsharded computation."""
    out = str(tmp_path / "r0.pt")
    mp.spawn(band_worker, args=(world, free_port(), out, total_rows), nprocs=world, join=True)
    got = torch.load(out)
    g = torch.Generator().manual_seed(11)
    H = 2 * total_rows
    x = torch.randn(world, 2, H, 6, generator=g)
    w1 = (torch.randn(4, 2, 3, 3, generator=g) * 0.4).requires_grad_(True)
    w2 = (torch.randn(1, 4, 3, 3, generator=g) * 0.4).requires_grad_(True)
    full = _band_net(x, w1, w2)
    assert torch.allclose(got["full"], full.detach(), atol=1e-6)
    loss = sum((full[i:i + 1] * torch.linspace(-1, 1, H * 6).reshape(1, 1, H, 6)).sum() + (full[i:i + 1] ** 2).mean()
               for i in range(world)) / world
    flat = torch.cat([t.reshape(-1) for t in torch.autograd.grad(loss, (w1, w2))])
    err = float((got["flat"] - flat).norm() / flat.norm())
    assert err < 1e-5, err


def heights_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from infinite_texture_gans_amd.dist import BandComm
        comm = BandComm(rank, world, dist.group.WORLD)
        t = torch.full((1, 1, 4 if rank == 0 else 6, 3), float(rank))       # rank 1 holds 6 rows but 4 are announced
        res = "ok"
        try:
            outs = comm.all_gather(t, heights=[4, 4])
            res = "ok %s" % [tuple(o.shape) for o in outs]
        except ValueError as e:
            res = "ValueError: %s" % e
        # both ranks are still in step: one more collective completes
        z = torch.ones(1)
        dist.all_reduce(z)
        torch.save((res, float(z)), out + str(rank))
    finally:
        dist.destroy_process_group()


def test_all_gather_with_wrong_static_heights_fails_on_the_bad_rank_without_hanging_the_others(tmp_path):
    """a rank whose band does not have the announced height used to raise BEFORE the collective and left the
    other ranks blocked in it.  Now it takes part with a correctly shaped stand-in and raises afterwards."""
    out = str(tmp_path / "h")
    mp.spawn(heights_worker, args=(2, free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + "0"), torch.load(out + "1")
    assert r0[0].startswith("ok") and r0[1] == 2.0
    assert r1[0].startswith("ValueError") and "announced as 4" in r1[0] and r1[1] == 2.0


def switches_worker(rank, world, port, out, differ):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.pop("ITG_BUCKETS", None)
    os.environ.pop("ITG_BUCKET_HEAD", None)
    if differ == "bucket_head_last_digit" and rank == 1:
        os.environ["ITG_BUCKET_HEAD"] = "0.21"       # a value whose hash differs by little (the float-variance test of r4 missed those)
    if differ == "bucket_head_malformed" and rank == 1:
        os.environ["ITG_BUCKET_HEAD"] = "0,2"        # float() raised on this rank BEFORE the all-reduces, the others hung
    if differ == "explicit_default" and rank == 1:
        os.environ["ITG_BUCKET_HEAD"] = "0.2"        # the default written out: NOT a disagreement (r4 raised a false mismatch)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from infinite_texture_gans_amd.dist import SyncGroup
        from infinite_texture_gans_amd.engine import Trainer

        class Stub:                                  # what Trainer._check_switches_agree reads of a trainer: RESOLVED decisions
            sync, sync_bn, device = SyncGroup(dist.group.WORLD), False, "cpu"
            overlap, sn_fused = True, False
            defer_reduce = differ == "defer_reduce_argument" and rank == 1     # --wgrad_reduce deferred on one rank only
        res = "agree"
        try:
            Trainer._check_switches_agree(Stub(), buckets=(differ == "buckets" and rank == 1))
        except RuntimeError as e:
            res = str(e)
        z = torch.ones(1)
        dist.all_reduce(z)                           # the ranks are still in step
        torch.save((res, float(z)), out + str(rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("differ", [None, "explicit_default", "buckets", "defer_reduce_argument", "bucket_head_last_digit",
                                    "bucket_head_malformed"])
def test_ranks_that_disagree_on_a_collective_switch_fail_together_instead_of_hanging(tmp_path, differ):
    """the bucketed exchange, the deferred reduce (constructor argument / --wgrad_reduce as much as the
    environment), sync-BN ... decide WHICH collectives a rank issues; if they differ between ranks the sequences diverge and
    the job hangs.  Trainer compares the RESOLVED decisions with two integer all-reduces (MIN, MAX) once they are all taken: a
    mismatch is a RuntimeError on EVERY rank, an explicitly written default is not a mismatch."""
    out = str(tmp_path / "sw")
    mp.spawn(switches_worker, args=(2, free_port(), out, differ), nprocs=2, join=True)
    r0, r1 = torch.load(out + "0"), torch.load(out + "1")
    assert r0[1] == 2.0 and r1[1] == 2.0
    word = {"buckets": "ITG_BUCKETS", "defer_reduce_argument": "--wgrad_reduce", "bucket_head_last_digit": "ITG_BUCKET_HEAD",
            "bucket_head_malformed": "ITG_BUCKET_HEAD"}.get(differ)
    if word:
        assert word in r0[0] and word in r1[0], (r0, r1)
    else:
        assert r0[0] == "agree" and r1[0] == "agree", (r0, r1)


def strips_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from infinite_texture_gans_amd.dist import RowHalo
        from infinite_texture_gans_amd import utils as U
        halo = RowHalo(rank, world, dist.group.WORLD)
        t_h, p, out_h, w = 7, 4, 26, 5                    # 7 patch rows of 4 pixels on 3 ranks (3 + 2 + 2), cropped to 26 rows
        rows = U.strip_rows(halo, t_h, p, out_h)
        lo, hi = rows[rank]
        full = torch.arange(1 * 2 * 28 * w, dtype=torch.float32).reshape(1, 2, 28, w)
        img = U.gather_strips(full[:, :, lo:hi].clone(), halo, rows)
        if rank == 0:
            torch.save({"rows": rows, "img": img}, out)
        else:
            assert img is None
    finally:
        dist.destroy_process_group()


def test_ragged_strips_of_a_row_sharded_image_are_gathered_as_tensors(tmp_path):
    """utils.gather_strips (test_sample.py on N ranks): rank r's strip of output rows - ragged bands, the last one cropped to the
    requested height - lands in rank 0's preallocated image by point-to-point tensor transfers (round 3: gather_object)."""
    out = str(tmp_path / "strips.pt")
    mp.spawn(strips_worker, args=(3, free_port(), out), nprocs=3, join=True)
    got = torch.load(out)
    assert got["rows"] == [(0, 12), (12, 20), (20, 26)]
    full = torch.arange(1 * 2 * 28 * 5, dtype=torch.float32).reshape(1, 2, 28, 5)
    assert torch.equal(got["img"], full[:, :, :26])


# ------------------------------------------------------------------------------- two-bucket gradient exchange
class _Flat:
    """What engine.GradExchange needs of engine.FlatParams: the flat gradient buffer."""

    def __init__(self, net_):
        self.numel = sum(p.numel() for p in net_.parameters())
        self.grad = torch.zeros(self.numel)
        o = 0
        self.views = []
        for p in net_.parameters():
            self.views.append(self.grad[o:o + p.numel()].view(p.shape))
            o += p.numel()


def _toy_d():
    import torch.nn as nn
    return nn.Sequential(nn.Sequential(nn.Conv2d(3, 4, 3), nn.LeakyReLU(0.2), nn.Conv2d(4, 8, 3), nn.LeakyReLU(0.2),
                                       nn.Conv2d(8, 16, 3), nn.LeakyReLU(0.2), nn.Conv2d(16, 1, 3)))


def test_bucket_split_points_follow_module_boundaries():
    """The head bucket ends at a unit boundary (a layer of an nn.Sequential / a top-level block) and holds >= 20 % of the
    parameters; config 1's models split as DESIGN.md section 5 says."""
    from infinite_texture_gans_amd.engine import GradExchange
    d = _toy_d()
    total = sum(p.numel() for p in d.parameters())
    sizes = [sum(p.numel() for p in m.parameters()) for m in d[0] if any(True for _ in m.parameters())]
    split = GradExchange._split_point(d, total)
    assert split in [sum(sizes[:k]) for k in range(1, len(sizes))] and split >= 0.2 * total
    assert split - sizes[[sum(sizes[:k]) for k in range(1, len(sizes))].index(split)] < 0.2 * total      # the first such boundary
    import bench
    from infinite_texture_gans_amd import utils as U
    args = U.prepare_parser().parse_args(bench.FLAGS)
    G, D = U.prepare_models(args, torch.device("cpu"))
    nG, nD = sum(p.numel() for p in G.parameters()), sum(p.numel() for p in D.parameters())
    sG, sD = GradExchange._split_point(G, nG), GradExchange._split_point(D, nD)
    assert sG == sum(p.numel() for p in G.start.parameters()) + sum(p.numel() for p in G.block1.parameters())
    assert 0.67 < sG / nG < 0.69 and 0.23 < sD / nD < 0.25


def exchange_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from infinite_texture_gans_amd.engine import GradExchange
        d = _toy_d()
        flat = _Flat(d)
        ex = GradExchange(flat, d, SyncGroup(dist.group.WORLD), "cpu")
        assert 0 < ex.split < flat.numel
        g = torch.Generator().manual_seed(100 + rank)
        res = []
        for it in range(2):
            flat.grad.copy_(torch.randn(flat.numel, generator=g))
            mine = flat.grad.clone()
            ex.arm()
            # backward order: last layer first; a tail layer's hook must not trigger, the first head layer's must
            ex._on_conv_backward((flat.views[-2], flat.views[-1]))
            assert not ex.early
            tail_before = flat.grad[ex.split:].clone()
            ex._on_conv_backward((flat.views[2], flat.views[3]))          # second conv: inside the head
            assert ex.early and not torch.equal(flat.grad[ex.split:], tail_before)
            assert torch.equal(flat.grad[:ex.split], mine[:ex.split])    # the head is still local
            ex._on_conv_backward((flat.views[0], flat.views[1]))          # later head layers: nothing more happens
            ex.finish()
            res.append((mine, flat.grad.clone()))
        torch.save(res, "%s.%d" % (out, rank))
    finally:
        dist.destroy_process_group()


def test_two_bucket_gradient_exchange_over_gloo_equals_the_average(tmp_path):
    out = str(tmp_path / "ex")
    mp.spawn(exchange_worker, args=(2, free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    for it in range(2):
        want = (r0[it][0] + r1[it][0]) / 2
        assert torch.allclose(r0[it][1], want, rtol=0, atol=1e-7) and torch.equal(r0[it][1], r1[it][1])


def test_band_latents_of_every_rank_are_slices_of_the_one_rank_draw():
    """config 5: each rank draws only the noise rows of its own band (per-patch-row seeds); the rows neighbouring bands
    share must be identical on both, i.e. every band is a slice of what a single rank draws with the same seed - also for
    the ragged split of 9 rows over 4 ranks."""
    from infinite_texture_gans_amd import utils as U
    from infinite_texture_gans_amd.dist import RowHalo

    class G:
        type_norm, n_layers_G = "SSM", 3

    t_h, t_w, b = 9, 4, 4
    zf, mf = U.band_latents(G, t_h, t_w, b, RowHalo(0, 1), "cpu", z_dim=5, map_dim=2, seed=11)
    assert zf.shape == (1, 5, t_h * b + 2, t_w * b + 2) and [m.shape[2] for m in mf] == [t_h * b * 2 ** i + 4 for i in range(3)]
    for world in (2, 4):
        for r in range(world):
            h = RowHalo(r, world)
            a, e = h.band(t_h)
            z, m = U.band_latents(G, t_h, t_w, b, h, "cpu", z_dim=5, map_dim=2, seed=11)
            assert torch.equal(z, zf[:, :, a * b:e * b + 2])
            for i in range(3):
                rr = b * 2 ** i
                assert torch.equal(m[i], mf[i][:, :, a * rr:e * rr + 4])
    # another seed is another image; the reference-RNG form cuts the given full-grid tensors
    z2, _ = U.band_latents(G, t_h, t_w, b, RowHalo(0, 1), "cpu", z_dim=5, map_dim=2, seed=12)
    assert not torch.equal(z2, zf)
    zc, mc = U.band_latents(G, t_h, t_w, b, RowHalo(1, 2), "cpu", z_full=zf, maps_full=mf)
    a, e = RowHalo(1, 2).band(t_h)
    assert torch.equal(zc, zf[:, :, a * b:e * b + 2]) and torch.equal(mc[2], mf[2][:, :, a * 16:e * 16 + 4])
