"""Train-step engine: flat parameter/gradient buffers, the fused Adam(+EMA) step, and the
D-step / G-step control flow of reference train.py:122-180.

MI355X-first choices (vs the reference's per-tensor torch.optim.Adam and nn.DataParallel):
  * every model's parameters are views into ONE contiguous fp32 buffer, their gradients views
    into another: zero_grad is one memset, Adam(+EMA) one kernel launch, and the data-parallel
    gradient exchange one RCCL all-reduce over xGMI per model (G 21 MB, D 11 MB);
  * one process per GPU; BatchNorm uses per-rank statistics like the reference's DataParallel replicas, or
    (sync_bn) statistics summed over ranks as fp64 sum/sumsq pairs, which gives an N-GPU step the
    single-process semantics of a batch N times larger;
  * the step is scheduled over four HIP streams on one GPU (D(real) beside the generator forward, weight
    gradients on two streams beside the input-gradient chain) and can be captured into a hipGraph.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .dist import SyncGroup, average_flat_gradient
from .models.layers import _BNParams


class FlatParams:
    """Re-homes a module's parameters (and .grad) into flat buffers."""

    def __init__(self, module):
        self.params = [p for p in module.parameters()]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.data = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        o = 0
        for p in self.params:
            k = p.numel()
            self.data[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + k].view(p.shape)
            p.grad = self.grad[o:o + k].view(p.shape)
            o += k
        self.numel = n

    def zero_grad(self):
        self.grad.zero_()
        # autograd must keep accumulating in place into the flat buffer
        for p, g in zip(self.params, self._grad_views()):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def _grad_views(self):
        o = 0
        for p in self.params:
            k = p.numel()
            yield self.grad[o:o + k].view(p.shape)
            o += k


class FlatAdam:
    """Adam (torch.optim.Adam semantics: bias correction, eps outside the sqrt/bc2 term, no weight
    decay) over a FlatParams, optionally maintaining an EMA copy of the parameters in the same launch."""

    def __init__(self, flat, lr=2e-4, betas=(0.0, 0.999), eps=1e-8, ema=None, ema_decay=0.999):
        self.flat, self.lr, self.betas, self.eps = flat, lr, (float(betas[0]), float(betas[1])), eps
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        self.ema, self.ema_decay = ema, ema_decay
        self.t = 0
        self.t_dev = torch.zeros((), device=flat.data.device, dtype=torch.int32)   # graph-replay-safe step count

    def step(self):
        self.t += 1
        self.t_dev.add_(1)
        ops.adam_ema_step(self.flat.data, self.flat.grad, self.m, self.v, self.ema, self.lr, self.betas[0],
                          self.betas[1], self.eps, self.t, self.ema_decay, step_dev=self.t_dev)


class GradExchange:
    """Data-parallel exchange of one model's flat gradient buffer in TWO buckets, the first one under the backward pass.

    Parameters sit in the flat buffer in module order and the backward pass runs through the modules in reverse, so
    the TAIL of the buffer is final first.  The buffer is split at a unit boundary (a top-level block; a layer of an
    nn.Sequential) such that the head holds >= ITG_BUCKET_HEAD (20 %) of the parameters: D = [D0 D1 D2 | D3 logit]
    (24 % | 76 %), G = [start block1 | block2.. final] (68 % | 32 %).  When the first conv of the head ENTERS its
    backward, every autograd node of the tail has been enqueued (its output gradient depends on all of them): the
    tail's all-reduce is then issued on a communication stream that waits for the backward stream and for the
    weight-gradient streams as they are at that moment, and runs beside the rest of the backward.  The head follows
    after the weight-gradient streams were joined.  RCCL over xGMI is ring / link bound (MI355X_MICROARCH.md): two
    collectives of several MB each, not one per layer.

    Only for models whose parameters are all written on the backward's own path (convs, BatchNorm, attention):
    the SSM generator's shared MLP collects gradients from every layer until the end and keeps the single bucket."""

    def __init__(self, flat, net, sync, device, two_buckets=True, issue_stream=None):
        """``issue_stream``: the stream the early bucket is exchanged on - one that is IDLE during the backward pass and
        has a hardware queue of its own (the step's D(real) branch stream); a fifth busy queue would cost more than the
        overlap gains (ops.concurrent_streams), and a collective parked in a queue that other kernels share holds them up
        until its inputs are ready.  The collective is a blocking (async_op=False) call made under that stream.  On which
        stream the RCCL kernel then runs depends on the torch build (2.10: the current stream, tools/probes/nccl_stream.py;
        older builds: the group's internal stream, ordered by events) - ordering is correct either way: the issuing stream
        has waited for the backward stream and the dirty weight-gradient streams, and finish() waits for the issuing
        stream.  UNVERIFIED on more than one RCCL rank (no multi-GPU box in this build's pool): the bucketed schedule is
        therefore opt-in (ITG_BUCKETS=1) for real multi-rank RCCL groups and the default there is one all-reduce per model
        after the joins."""
        self.flat, self.sync = flat, sync
        self.world = sync.world
        self.split = self._split_point(net, flat.numel) if two_buckets else 0
        self.cuda = torch.device(device).type == "cuda"          # CPU tensors (gloo protocol tests): same buckets, no streams
        self.comm = (issue_stream or torch.cuda.Stream(device=device)) if self.split and self.cuda else None
        self.early = False
        self._armed = False

    @staticmethod
    def _split_point(net, total):
        frac = float(os.environ.get("ITG_BUCKET_HEAD", "0.2"))
        if frac <= 0 or any(True for _ in net.parameters(recurse=False)):
            return 0
        units = []
        for name, c in net.named_children():
            if isinstance(c, nn.Sequential):
                units.extend((name, u) for u in c.children())
            else:
                units.append((name, c))
        # The hook fires when the first conv of the head enters its backward, which only proves that the tail is complete
        # if every unit behind the split also EXECUTES behind it.  The generator registers `bn` and `attention` after all
        # blocks although attention runs between block3 and block4: a split that would put such an out-of-order unit in
        # the tail (or the blocks it precedes in the head) is refused - the scan stops at the last block that executes
        # before the first out-of-order unit.
        units = [(n, u) for n, u in units if any(True for _ in u.parameters())]
        order = getattr(net, "execution_order", None)
        limit = len(units)
        if order is not None:
            rank_of = {n: i for i, n in enumerate(order)}
            pos = [rank_of.get(n, len(order)) for n, _ in units]
            for i in range(len(units)):
                if any(pos[j] < pos[i] for j in range(i + 1, len(units))):      # a later-registered unit runs earlier
                    limit = i
                    break
        seen = 0
        for i, (_, u) in enumerate(units):
            if i >= limit:
                return 0
            seen += sum(p.numel() for p in u.parameters())
            if seen >= frac * total:
                return seen if seen < total else 0
        return 0

    def arm(self):
        """Call right before the backward pass whose gradients are exchanged (the LAST pass that accumulates into them)."""
        if self.split:
            self._armed = True
            ops.BACKWARD_ENTRY_HOOK = self._on_conv_backward

    def _on_conv_backward(self, sinks):
        if not self._armed:
            return
        sink = sinks[0] if sinks[0] is not None else sinks[1]
        off = (sink.data_ptr() - self.flat.grad.data_ptr()) // 4
        if not 0 <= off < self.split:
            return                                       # a layer of the tail, or of the other model
        self._armed = False
        if self.cuda:
            cur = torch.cuda.current_stream()
            self.comm.wait_stream(cur)
            for s_ in ops._wgrad_dirty:                  # the tail's weight gradients (and whatever else is queued there)
                self.comm.wait_stream(s_)
            with torch.cuda.stream(self.comm):
                self.sync.all_reduce(self.flat.grad[self.split:])
        else:
            self.sync.all_reduce(self.flat.grad[self.split:])
        self.early = True

    def finish(self):
        """After the weight-gradient streams were joined: exchange what is left, wait for the early bucket, average."""
        ops.BACKWARD_ENTRY_HOOK = None
        self._armed = False
        g = self.flat.grad
        if self.early:
            self.sync.all_reduce(g[:self.split])
            if self.cuda:
                torch.cuda.current_stream().wait_stream(self.comm)
            self.early = False
        else:
            self.sync.all_reduce(g)
        g.mul_(1.0 / self.world)


class PackSet:
    """The packed filter panels of every conv of a model, refreshed with ONE launch whenever the
    optimizer has changed the weights (instead of two pack launches per conv call).  Layers find their
    panels in ``_packed``; ``release()`` returns them to per-call packing (weights edited by hand)."""

    def __init__(self, net):
        from .models.layers import _ConvParams
        self.layers = [m for m in net.modules() if isinstance(m, _ConvParams)]
        self.jobs = [j for m in self.layers for j in m.pack_jobs()]      # keeps the weight / panel tensors alive
        self.tables = ops.pack_tables(self.jobs, self.jobs[0][0].device) if self.jobs else []

    def repack(self):
        ops.pack_multi(self.tables)

    def release(self):
        for m in self.layers:
            m._packed = None


class Trainer:
    """One G+D iteration of reference train.py:122-180 on the HIP kernels."""

    def __init__(self, netG, netD, args, device, netG_ema=None, dist_group=None, sync_bn=None, defer_reduce=None):
        """``dist_group``: data parallel over its ranks (one flat gradient all-reduce per model and step).
        ``sync_bn``: False (default, ITG_SYNC_BN=0) = every rank normalises with the statistics of its own batch,
        which is what the reference's nn.DataParallel does (train.py:74-77); True = BatchNorm sums are all-reduced
        so that N ranks reproduce the single-process step at batch 8N (26 small collectives per step)."""
        self.netG, self.netD, self.args, self.device = netG, netD, args, device
        self.world = 1
        self.sync = None
        if sync_bn is None:
            sync_bn = os.environ.get("ITG_SYNC_BN", "1" if getattr(args, "sync_bn", False) else "0") == "1"
        self.sync_bn = bool(sync_bn) and dist_group is not None
        if dist_group is not None:
            self.sync = sync = SyncGroup(dist_group)
            self.world = sync.world
            if self.sync_bn:
                netG.set_sync(sync)
                for m in netD.modules():
                    if isinstance(m, _BNParams):
                        m.sync = sync
        self.flatG, self.flatD = FlatParams(netG), FlatParams(netD)
        from .models.layers import _ConvParams
        for net in (netG, netD):        # backward kernels accumulate straight into the flat .grad buffers
            for m in net.modules():
                if isinstance(m, (_ConvParams, _BNParams)):
                    m.grad_sinks = True
        self.netG_ema, self.flatE = netG_ema, None
        if netG_ema is not None:
            self.flatE = FlatParams(netG_ema)
            self.flatE.data.copy_(self.flatG.data)
            for (_, b), (_, e) in zip(netG.named_buffers(), netG_ema.named_buffers()):
                e.copy_(b)
        b1, b2 = float(args.beta1), float(args.beta2)
        self.optD = FlatAdam(self.flatD, args.lr_D, (b1, b2))
        self.optG = FlatAdam(self.flatG, args.lr_G, (b1, b2), ema=self.flatE.data if self.flatE else None,
                             ema_decay=args.ema_decay)
        self.label_t = 0.9 if args.smooth else 1.0
        self._one = torch.ones((), device=device)               # the seed of every backward pass (no ones_like fill per pass)
        self.hinge = getattr(args, "loss", "standard") == "hinge"
        # the loss heads read D's logit map in the grid layout its last conv wrote: loss + derivative in one launch (ops.logit_loss)
        self.fused_loss = True
        self.packG, self.packD = PackSet(netG), PackSet(netD)
        # dx buffers of the generator's replicate-padded convs, kept across steps (ops.begin_frames)
        self._frames = {}
        self.repack()
        self.arena = ops.ZeroArena(device)
        # stream overlap (D(real) beside the generator forward, weight gradients beside the input-gradient chain).
        # Off by default when BatchNorm statistics are all-reduced: those 26 latency-critical collectives then
        # queue behind the kernels of the side streams (one-rank RCCL rehearsal: 12.07 ms with vs 12.13 ms without
        # overlap, against 10.86 ms without the collectives).  The two gradient all-reduces come after the joins.
        self.overlap = os.environ.get("ITG_OVERLAP", "0" if self.sync_bn else "1") == "1"
        self.side, self._wstream, self.wstream = None, None, None
        # D(real)'s weight gradients may leave their branch stream for the weight-gradient streams (a fork of a fork;
        # also under hipGraph capture - what used to crash there was a wait for an idle forked stream, see
        # ops.wgrad_streams_join)
        self.nested_fork = os.environ.get("ITG_NESTED_FORK", "1") == "1"
        # weight gradients in the deferred form: slabs per layer on the weight-gradient streams, ONE reduce launch per
        # backward pass at the join (ops.flush_deferred)
        # (ITG_DEFER_REDUCE=1; off by default.  Measured on MI355X, config 1: 353 -> 298 launches and 0.45 ms less kernel time per
        # step, but 765 vs 780 crops/s: the per-layer reduce launches run on the weight-gradient streams in the shadow of the
        # input-gradient chain, which is the critical path; the single reduce at the join - ~50 MB of slabs per wide layer,
        # bandwidth-bound at ~150 us per pass - is ON that path.)
        # (``defer_reduce`` argument: train.py / bench.py decide per workload - on for the launch-bound config 3 under graph replay)
        self.defer_reduce = (os.environ.get("ITG_DEFER_REDUCE", "0") == "1") if defer_reduce is None else bool(defer_reduce)
        self._defer = []
        # spectrally normalised layers that are not deferred finish their weight gradient through the job reduce (dot in the
        # reduce launch + one apply launch): -10 launches; pays where the step is launch-bound (config 3 under replay: 2 473 ->
        # 2 497 crops/s), costs 0.2 % on config 1 - so it follows the deferred reduce unless ITG_SN_FUSED_REDUCE says otherwise
        self.sn_fused = (os.environ["ITG_SN_FUSED_REDUCE"] == "1") if "ITG_SN_FUSED_REDUCE" in os.environ else self.defer_reduce
        self.set_overlap(self.overlap)
        from .dist import _active
        # two-bucket exchange: default on for the rehearsal backends (gloo / one-rank RCCL, where it is tested), opt-in on a
        # real multi-rank RCCL group until it has run there once
        multi_rccl = False
        if dist_group is not None and self.world > 1 and torch.device(device).type == "cuda":
            import torch.distributed as _dist
            multi_rccl = _dist.get_backend(dist_group) == "nccl"      # the backend of the group that was passed, not the environment
        buckets = os.environ.get("ITG_BUCKETS", "0" if multi_rccl else "1") == "1"
        self._check_switches_agree(buckets)
        if self.sync is not None and _active(self.sync) and buckets:
            # the early buckets travel on the D(real) branch stream: idle during both backward passes that are exchanged
            issue = self.side if self.overlap else None
            early = not self.defer_reduce      # deferred reduces finish the tail's gradients only at the join: single bucket
            self._exchange = {
                id(self.flatD): GradExchange(self.flatD, netD, self.sync, device, issue_stream=issue, two_buckets=early),
                id(self.flatG): GradExchange(self.flatG, netG, self.sync, device, issue_stream=issue,
                                             two_buckets=early and getattr(netG, "type_norm", "BN") == "BN"),
            }

        self._warm_collectives()

    def _check_switches_agree(self, buckets):
        """The RESOLVED per-process decisions about which collectives a rank issues (bucketed exchange and its head share,
        warm-up, sync-BN, deferred reduces - constructor argument, --wgrad_reduce or environment -, the spectral-norm reduce that
        follows them, stream overlap) must be equal on all ranks, or the collective sequences diverge and the job hangs
        (ADVICE r3 / r4).  Called once every decision is taken and before the first conditional collective; two unconditional
        integer all-reduces (MIN, MAX) and exact equality: a mismatch raises on every rank."""
        if self.sync is None or self.sync.world <= 1:
            return
        names = ("bucketed gradient exchange (ITG_BUCKETS)", "head bucket share (ITG_BUCKET_HEAD)", "collective warm-up (ITG_WARM_COLLECTIVES)",
                 "sync-BN (--sync_bn / ITG_SYNC_BN)", "deferred weight-gradient reduce (--wgrad_reduce / ITG_DEFER_REDUCE)",
                 "fused spectral-norm reduce (ITG_SN_FUSED_REDUCE)", "stream overlap (ITG_OVERLAP)")
        try:
            head = int(round(float(os.environ.get("ITG_BUCKET_HEAD", "0.2")) * 1e6))
        except ValueError:
            head = -1          # a malformed value on this rank must still reach both all-reduces: the mismatch raises on every rank
        mine = [int(bool(buckets)), head,
                int(os.environ.get("ITG_WARM_COLLECTIVES", "1") == "1"), int(self.sync_bn), int(self.defer_reduce), int(self.sn_fused),
                int(self.overlap)]
        dev = self.device if torch.device(self.device).type == "cuda" else "cpu"
        lo = torch.tensor(mine, dtype=torch.int64, device=dev)
        hi = lo.clone()
        self.sync.dist.all_reduce(lo, op=self.sync.dist.ReduceOp.MIN, group=self.sync.group)
        self.sync.dist.all_reduce(hi, op=self.sync.dist.ReduceOp.MAX, group=self.sync.group)
        bad = [n for n, a, b in zip(names, lo.tolist(), hi.tolist()) if a != b]
        if not bad and head < 0:
            bad = ["head bucket share (ITG_BUCKET_HEAD is not a number)"]
        if bad:
            raise RuntimeError("the ranks of this job disagree on %s: set it identically on every rank" % ", ".join(bad))

    def set_defer_reduce(self, on):
        """Switch the deferred weight-gradient reduce between steps (train.py --launch_mode auto decides it with the launch
        mode, after its probe).  Single-rank engines only: the gradient exchange fixed its bucket plan at construction."""
        if self._exchange:
            raise RuntimeError("the deferred reduce of a data-parallel engine is fixed at construction")
        self.defer_reduce = bool(on)
        if "ITG_SN_FUSED_REDUCE" not in os.environ:
            self.sn_fused = self.defer_reduce

    def _warm_collectives(self):
        """RCCL sets up channels, proxy threads and per-size algorithm state lazily inside the first collectives of a process
        (hundreds of milliseconds): run the step's all-reduces once on zeroed buffers of the real sizes here, so that the
        first training steps - and a short benchmark - do not pay for it.  (ITG_WARM_COLLECTIVES=0 skips it.)"""
        from .dist import _active
        if (self.sync is None or not _active(self.sync) or torch.device(self.device).type != "cuda"
                or os.environ.get("ITG_WARM_COLLECTIVES", "1") != "1"):
            return
        for flat in (self.flatD, self.flatG):
            buf = torch.zeros_like(flat.grad)
            ex = self._exchange.get(id(flat)) if self._exchange else None
            parts = [buf] if ex is None or not ex.split else [buf[ex.split:], buf[:ex.split]]
            for _ in range(3):
                for part in parts:
                    self.sync.all_reduce(part)
        stat = torch.zeros(2 * 512, device=self.device, dtype=torch.float64)       # sync-BN statistics size class
        self.sync.all_reduce(stat)
        torch.cuda.synchronize()

    def set_overlap(self, on):
        """Stream overlap on / off (off: every kernel runs alone on the current stream, e.g. to time it).  The branch
        stream and the weight-gradient streams are chosen by measurement so that each sits on its own hardware queue
        (ops.concurrent_streams): four busy queues in all, which is what the chip runs well."""
        self.overlap = bool(on)
        if on and self.side is None:
            nws = max(1, min(2, int(os.environ.get("ITG_WGRAD_STREAMS", "2"))))
            picked = ops.concurrent_streams(self.device, 1 + nws)
            self.side, self._wstream = picked[0], picked[1:]
        self.wstream = self._wstream if on else None

    def repack(self):
        """Refresh the packed filter panels (call after changing weights outside the optimizer steps)."""
        self.packG.repack()
        self.packD.repack()

    # ---- loss heads
    def _d_loss(self, logit, real):
        if isinstance(logit, ops.GT):        # the logit map as D's last conv left it: loss + derivative in one launch
            return ops.logit_loss(logit, ("d_real" if real else "d_fake") if self.hinge else "bce", self.label_t if real else 0.0)
        if self.hinge:
            return ops.hinge(logit, "d_real" if real else "d_fake")
        return ops.bce_with_logits(logit, self.label_t if real else 0.0)

    def _g_loss(self, logit):
        if isinstance(logit, ops.GT):
            return ops.logit_loss(logit, "g" if self.hinge else "bce", self.label_t)
        return ops.hinge(logit, "g") if self.hinge else ops.bce_with_logits(logit, self.label_t)

    def _allreduce(self, flat):
        """Data-parallel gradient exchange (losses are per-rank means over per-rank batches, so the global-batch
        gradient is the average over ranks): two buckets per model, the first under the backward (GradExchange)."""
        ex = self._exchange.get(id(flat)) if self._exchange else None
        if ex is not None:
            ex.finish()
        else:
            average_flat_gradient(flat.grad, self.sync)

    def _arm_exchange(self, flat):
        ex = self._exchange.get(id(flat)) if self._exchange else None
        if ex is not None:
            ex.arm()

    _exchange = None

    def sample_fake(self, z, maps):
        return self.netG.forward_grid(z, maps, "1st_row_1st_col")

    record = None       # a list: every logit map the loss heads see is appended to it (parity tests)
    marks = None        # a list: (name, event) pairs recorded on the main stream at the phase boundaries (tools/region_times.py)

    def _mark(self, name):
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, ev))

    def _d_logits(self, fake):
        """D on the generator's output: a patch grid (consumed in place, no merge copy) or whole NCHW images.  The logit map
        stays in the grid layout (ops.GT) for the fused loss head (`Trainer.fused_loss = False`: NCHW logits and the generic heads)."""
        grid = hasattr(self.netD, "forward_grid")
        if isinstance(fake, ops.GT):
            lg = self.netD.forward_grid(fake)
        elif grid and self.fused_loss:
            lg = self.netD.forward_grid(ops.to_grid(fake, 1, 1, merged=True))
        else:
            lg = self.netD(fake)
        if isinstance(lg, ops.GT) and not (self.fused_loss and lg.c == 1):
            lg = ops.to_nchw(lg)
        if self.record is not None:
            self.record.append((ops.to_nchw(lg) if isinstance(lg, ops.GT) else lg).detach())
        return lg

    def _d_real_logits(self, real_x):
        return self._d_logits(real_x)

    def step(self, real_x, z, maps=None, next_real=None):
        """real_x: (B,3,crop,crop) on the device; z/maps: latents (see utils.sample_latents_train), or LISTS of
        them: ``--disc_iters`` = len(z) discriminator updates on the same real batch with fresh latents each, then
        ONE generator update on the last fake batch (reference train.py:124-169).
        ``next_real``: the NEXT iteration's real batch, if the caller already has it.  D(real) forward+backward of the
        next iteration only depends on D's weights after this iteration's Adam(D): it is then issued on the side stream
        beside this iteration's generator backward (a chain of small latency-bound launches) instead of beside the next
        generator forward; the next ``step`` must be called with that very tensor as ``real_x``.  Same arithmetic, same
        spectral-norm power-iteration order (it is issued after the G step's D forward).
        Returns (d_loss_real, d_loss_fake, g_loss) of the last D iteration as 0-dim device tensors (no host sync);
        ``self.d_losses`` holds the (real, fake) pair of every D iteration."""
        self.arena.reset()                                      # BatchNorm statistics scratch of this iteration
        ops.ARENA = self.arena
        ops.WGRAD_STREAM = self.wstream
        ops.WGRAD_DEFER = self._defer if self.defer_reduce else None
        ops.UNIT_GRAD = self._one
        sn_keep, ops.SN_FUSED_REDUCE = ops.SN_FUSED_REDUCE, self.sn_fused
        try:
            zs = list(z) if isinstance(z, (list, tuple)) else [z]
            ms = list(maps) if isinstance(maps, (list, tuple)) and isinstance(z, (list, tuple)) else [maps] * len(zs)
            if len(ms) != len(zs):
                raise ValueError("step(): %d latents but %d map sets" % (len(zs), len(ms)))
            self.d_losses = []
            for zi, mi in zip(zs, ms):
                d_real, d_fake, fake = self.d_step(real_x, zi, mi)
                self.d_losses.append((d_real, d_fake))
            g_loss = self.g_step(fake, next_real)
            return d_real, d_fake, g_loss
        finally:                                                # never leave the process-wide hooks set behind an exception
            ops.ARENA = None
            ops.WGRAD_STREAM = None
            ops.WGRAD_DEFER = None
            ops.UNIT_GRAD = None
            ops.SN_FUSED_REDUCE = sn_keep
            ops.BACKWARD_ENTRY_HOOK = None
            if self.wstream is not None and not torch.cuda.is_current_stream_capturing():
                ops.WGRAD_KEEPALIVE.clear()

    def d_step(self, real_x, z, maps=None):
        """One discriminator update (reference train.py:126-153): D(real) fwd+bwd, G forward (graph kept for the
        generator update), D(fake.detach()) fwd+bwd, Adam(D).  -> (d_loss_real, d_loss_fake, fake)."""
        netG, netD = self.netG, self.netD
        if self.wstream is not None and not torch.cuda.is_current_stream_capturing():
            # not needed for ordering (every fork waits for a fresh event) but measured worth 4-7 % of the eager step:
            # it keeps the weight-gradient queues from running ahead into the next step's allocations.  Never under
            # capture: an idle forked stream that something later waits for is the EndCapture crash (ops.wgrad_streams_join)
            for ws in self.wstream:
                ws.wait_stream(torch.cuda.current_stream())
        self._mark("start")
        pending, self._pending = self._pending, None
        if pending is not None and pending[0] is not real_x:
            raise RuntimeError("step(next_real=...) of the previous iteration announced another real batch than this one")
        if pending is not None:
            # D(real) of this batch already ran (or is running) on the side stream, gradients zeroed ahead of it
            d_real = pending[1]
            fake = self.sample_fake(z, maps)
            self._mark("G forward (beside D(real) fwd+bwd)")
            torch.cuda.current_stream().wait_stream(self.side)
        else:
            self.flatD.zero_grad()
        if pending is not None:
            pass
        elif self.overlap:
            # D(real) forward+backward and the generator forward are independent and neither fills the chip
            # on its own (small grids, latency-bound normalisation kernels): run them on two HIP streams
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)
            keep = ops.WGRAD_STREAM
            if not self.nested_fork or (self.defer_reduce and torch.cuda.is_current_stream_capturing()):
                # D(real)'s weight gradients stay on the branch stream (under hipGraph capture also when the reduces are
                # deferred: joining the weight-gradient streams INTO the branch stream and forking onto them again later
                # is another shape of the ROCm 7.2 hipStreamEndCapture crash, tools/capture_nested_fork.py)
                ops.WGRAD_STREAM = None
            with torch.cuda.stream(self.side):
                d_real = self._d_loss(self._d_real_logits(real_x), True)
                d_real.backward(self._one)
                self._finish_d_real()
            ops.WGRAD_STREAM = keep
            fake = self.sample_fake(z, maps)
            self._mark("G forward (beside D(real) fwd+bwd)")
            ops.join_stream(self.side, "D(real) branch join", main)      # D(fake) continues D's spectral-norm state and .grad
        else:
            d_real = self._d_loss(self._d_real_logits(real_x), True)
            d_real.backward(self._one)
            self._finish_d_real()
            fake = self.sample_fake(z, maps)                   # GT patches, graph kept for the G step
        d_fake = self._d_loss(self._d_logits(fake.detach()), False)
        self._mark("D(fake) forward")
        self._arm_exchange(self.flatD)                   # D(real)'s pass has accumulated (or is ordered before this one per layer)
        d_fake.backward(self._one)
        self._mark("D(fake) backward: dgrad chain (wgrads on side streams)")
        self._join()
        self._allreduce(self.flatD)
        self.optD.step()
        self.packD.repack()
        self._mark("join wgrads + Adam(D) + repack")
        return d_real.detach(), d_fake.detach(), fake

    def g_step(self, fake, next_real=None):
        """The generator update on ``fake`` (reference train.py:161-169, + EMA :176-180).  D's weight gradients of
        this pass are never read (zeroed at the next D step), so they are not computed."""
        self.flatG.zero_grad()
        for p in self.flatD.params:
            p.requires_grad_(False)
        try:
            g_loss = self._g_loss(self._d_logits(fake))
            self._mark("G step: D forward")
        finally:
            for p in self.flatD.params:
                p.requires_grad_(True)
        # D(real) of the next iteration beside the generator backward.  (Issuing it after the backward instead, so that it
        # would run beside the all-reduce of G's head bucket, costs 5 % on one rank: 744 vs 778 crops/s - not done.)
        if next_real is not None and self._can_prefetch():
            self._prefetch_d_real(next_real)
        self._arm_exchange(self.flatG)
        if self._frames is not None:
            ops.begin_frames(self._frames)       # one launch zeroes the frames of all replicate-padded dx buffers
        try:
            g_loss.backward(self._one)
        finally:
            ops.end_frames()
        self._mark("G step: backward through D and G (G wgrads on side streams)")
        self._join()
        self._allreduce(self.flatG)
        self.optG.step()                                        # + EMA of the parameters (train.py:176-180)
        self.packG.repack()
        if self.netG_ema is not None:
            self._ema_buffers()
        self._mark("join wgrads + Adam(G) + repack")
        return g_loss.detach()

    def _can_prefetch(self):
        """D(real) of the next iteration may run ahead: only with the stream overlap, outside hipGraph capture, and for a
        discriminator without BatchNorm (its statistics scratch is reset at the start of every step)."""
        return (self.overlap and not torch.cuda.is_current_stream_capturing()
                and not any(isinstance(m, _BNParams) for m in self.netD.modules()))

    def _prefetch_d_real(self, next_real):
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)                      # after Adam(D), the repack and the G step's D forward
        keep, ops.WGRAD_STREAM = ops.WGRAD_STREAM, None  # its weight gradients stay on the branch stream
        arena, ops.ARENA = ops.ARENA, None               # it outlives this step: no scratch from the arena the next step zeroes
        try:
            with torch.cuda.stream(self.side):
                self.flatD.zero_grad()                   # the gradients Adam(D) of this iteration consumed
                d_real = self._d_loss(self._d_real_logits(next_real), True)
                d_real.backward(self._one)
                self._finish_d_real()
        finally:
            ops.WGRAD_STREAM = keep
            ops.ARENA = arena
        self._pending = (next_real, d_real.detach())

    _pending = None

    def _join(self):
        """The weight-gradient streams have to drain before gradients are exchanged / consumed by Adam; the queued reduces of
        this backward pass run behind them, in one launch."""
        if self.wstream is not None:
            ops.flush_deferred(per_stream=True)      # each weight-gradient stream reduces its own layers, then is joined
            ops.wgrad_streams_join()
            ops.flush_deferred()                     # (ops.DEFER_PER_STREAM = False: the single reduce behind the join)
            ops.WGRAD_KEEPALIVE.clear()
        else:
            ops.flush_deferred()

    def _finish_d_real(self):
        """End of D(real)'s backward on whatever stream it ran: its queued weight gradients are reduced THERE (D(fake)'s
        pass accumulates into the same .grad afterwards and is ordered behind this stream)."""
        if ops.WGRAD_DEFER:
            ops.flush_deferred(per_stream=True)
            ops.wgrad_streams_join()
            ops.flush_deferred()

    # ---- hipGraph: the whole iteration (~600 launches) as one graph replay
    def capture(self, real_x, z, maps=None, warmup=2):
        """Record one iteration into a hipGraph.  ``warmup`` ordinary (training!) iterations run first on a
        side stream so that every lazily created buffer exists; inputs are copied into static buffers on
        each :meth:`step_graphed`.  Everything the step needs is stream-ordered device work: weights, Adam
        state and step counters, BN/SN buffers all live in device memory, nothing is read back."""
        self._g_real, self._g_z = real_x.clone(), z.clone()
        self._g_maps = None if maps is None or maps[0] is None else [m.clone() for m in maps]
        keep_fork = self.nested_fork
        if "ITG_NESTED_FORK" not in os.environ:
            # a replayed graph places its branches itself: D(real)'s weight gradients leaving their branch stream help the
            # eager queues (+1 %) and cost the replay 5 % (config 1: 1 047 vs 1 101 crops/s, the eager number; config 3 neutral)
            self.nested_fork = False
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step(self._g_real, self._g_z, self._g_maps)
        torch.cuda.current_stream().wait_stream(side)
        # The recording itself, by hand instead of `with torch.cuda.graph(...)`: between the step and capture_end the capture
        # rule is checked (ops.capture_rule: on ROCm 7.2 hipStreamEndCapture segfaults when a forked stream without a node of
        # the capture was waited for; such a wait is skipped while recording and reported here as a Python error, after the
        # capture has been closed cleanly).
        import gc
        from . import _lib
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        self.graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        del ops.CAPTURE_ERRORS[:]
        t_before = (self.optD.t, self.optG.t)      # recording does not train: the host-side step counters are put back
        try:
            with torch.cuda.stream(cap):
                _lib.CAPTURE_LOG = {cap.cuda_stream}
                self.graph.capture_begin()
                try:
                    self._g_out = self.step(self._g_real, self._g_z, self._g_maps)
                finally:
                    _lib.CAPTURE_LOG = None
                    self.graph.capture_end()
            torch.cuda.current_stream().wait_stream(cap)
            if ops.CAPTURE_ERRORS:
                errs = list(ops.CAPTURE_ERRORS)
                del ops.CAPTURE_ERRORS[:]
                raise RuntimeError("Trainer.capture: the step's stream schedule breaks the capture rule (the recorded graph would have "
                                   "crashed hipStreamEndCapture and was discarded): " + "; ".join(errs))
        except BaseException:
            # whatever failed, the engine stays usable for eager steps (train.py --launch_mode auto falls back to them)
            self.graph, self.nested_fork = None, keep_fork
            self.optD.t, self.optG.t = t_before
            raise
        self.optD.t, self.optG.t = t_before
        return self

    def step_graphed(self, real_x, z, maps=None):
        self._g_real.copy_(real_x, non_blocking=True)
        self._g_z.copy_(z, non_blocking=True)
        if self._g_maps is not None:
            for d, s_ in zip(self._g_maps, maps):
                d.copy_(s_, non_blocking=True)
        self.graph.replay()
        self.optD.t += 1
        self.optG.t += 1
        return self._g_out

    def _ema_buffers(self):
        d = self.args.ema_decay
        for (_, b), (_, e) in zip(self.netG.named_buffers(), self.netG_ema.named_buffers()):
            if b.dtype == torch.float32:
                ops.axpby(e, b, d, 1.0 - d, out=e)
            else:   # int64 num_batches_tracked: float arithmetic, truncated on copy (train.py:178)
                e.copy_(e * d + b * (1 - d))


class BandTrainer(Trainer):
    """The same iteration with the generator's patch grid sharded by patch ROWS over ranks (one fake
    image larger than one GPU wants to hold: BASELINE config 4).

    Rank r generates a band of patch rows of EVERY fake image: each 3x3 conv exchanges one pixel row
    with each neighbour (forward) and the halo gradients travel back the same way (backward);
    BatchNorm statistics are summed over bands.  In front of the discriminator the bands are gathered
    into whole images and D runs data-parallel over images (rank r scores images r*n/W..), so the
    gradient of a band is the sum over the ranks that scored its images.  Real crops are sharded over
    ranks as in plain data parallelism.  Results equal the single-process step on the whole batch.

    ``z`` handed to :meth:`step` is the FULL merged latent, identical on all ranks (same seed); SSM generators also take the
    MERGED noise maps (sample_fake).  Generators with per-patch attention re-grid their band into patches around the
    attention layer (ResidualPatchGenerator.band_layout)."""

    def __init__(self, netG, netD, args, device, comm, netG_ema=None):
        from .models.layers import LocalPadder, _ConvParams
        for m in netG.modules():          # bands carry explicit halo rows (pad_h = 0): the generator's convs keep the direct
            if isinstance(m, _ConvParams):     # kernels and their panels (decided before the panels are allocated)
                m.wino_ok = False
        super().__init__(netG, netD, args, device, netG_ema=netG_ema)
        if netG.padding_mode != 'local':
            raise ValueError("row sharding is defined for padding_mode='local'")

        # the step itself is Trainer's (real_x: this rank's shard of real crops; z: the FULL merged latent); with more
        # than one rank the generator's forward/backward carries halo exchanges and band-wide BatchNorm sums, which
        # must not queue behind side-stream kernels: stream overlap only for a single rank
        self.comm, self.sync, self.world = comm, comm, comm.world
        self.set_overlap(os.environ.get("ITG_OVERLAP", "1" if comm.world == 1 else "0") == "1")
        self.total_rows = netG.num_patches_h
        self.band = comm.band(self.total_rows)
        netG.band_layout = (self.band[1] - self.band[0], netG.num_patches_w)     # per-patch attention runs on the band's patch grid
        netG.set_sync(comm.band_sync(self.total_rows))
        for m in netD.modules():
            if isinstance(m, _BNParams):
                m.sync = comm
        for m in netG.modules():                         # the band is one "patch" in image layout
            if isinstance(m, LocalPadder):
                m.pin(1, 1, netG.outer_padding)
                m.halo = comm if m.merge_patches_into_image else None

    _heights = None

    def _mine(self, n):
        if n % self.world:
            raise ValueError("batch of %d images does not split evenly over %d ranks" % (n, self.world))
        k = n // self.world
        return slice(self.comm.rank * k, (self.comm.rank + 1) * k)

    def sample_fake(self, z, maps=None):
        """-> this rank's share of whole fake images (NCHW), differentiable w.r.t. every band.
        SSM generators: ``maps`` are the MERGED noise maps of the whole grid, one per layer, (N, map_dim, nph*r + 4, npw*r + 4)
        as the reference draws them (utils.py:506-519) - the per-patch crops the reference feeds its generator are windows of
        these, and the two valid 3x3 convs of the modulation MLP commute with the cropping, so a band runs them on its rows
        of the merged map (rows a*r .. b*r + 4)."""
        a, b = self.band
        r = self.netG.base_res
        band_maps = None
        if self.netG.type_norm == 'SSM':
            if maps is None or maps[0] is None:
                raise ValueError("band training of an SSM generator needs the merged noise maps")
            band_maps = []
            for i, m in enumerate(maps):
                ri = (2 ** i) * r
                if m.shape[-2] != self.total_rows * ri + 4:
                    raise ValueError("map %d has %d rows: band training takes the MERGED maps (%d rows), not per-patch crops"
                                     % (i, m.shape[-2], self.total_rows * ri + 4))
                band_maps.append(m[:, :, a * ri:b * ri + 4, :].contiguous())
        band = ops.to_nchw(self.netG.forward_grid(z[:, :, a * r:b * r + 2, :].contiguous(), band_maps, "1st_row_1st_col"),
                           merged=True)
        if self._heights is None:      # static for the model: rows per rank x patch height
            self._heights = self.comm.band_heights(self.total_rows, band.shape[-2] // (b - a))
        full = ops.gather_rows(band, self.comm, self._heights)
        return full[self._mine(full.shape[0])].contiguous()

