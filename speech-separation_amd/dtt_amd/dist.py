"""Data-parallel gradient exchange for one process per GPU (RCCL over xGMI).

The reference has no comm code of its own: HF accelerate wraps the model in
torch DDP (src/utils/init_utils.py:83-96, train.py:87,101).  This is a
from-scratch replacement sized for MI355X: gradients live in a few large flat
fp32 buckets (one multi-tensor gather per bucket) laid out in reverse-forward order (decoder ->
latent -> encoder, the order backward produces them); a bucket's all-reduce is
issued asynchronously the moment its last gradient has been accumulated, so
the collective overlaps the rest of backward.  xGMI is point-to-point, so few
large messages beat many small ones: the default bucket is 32 MiB (a quarter of the gradients for small models).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradBucketReducer:
    """Gradients are produced by autograd as free tensors (``zero_grad`` sets them to None, so the first
    accumulation is a move, not an add kernel per parameter).  When the last gradient of a bucket has
    arrived they are gathered into the bucket's flat fp32 buffer by ONE multi-tensor copy, ``p.grad`` is
    re-pointed at the views, and the bucket's all-reduce is issued asynchronously.  With one process
    (world size 1) nothing is exchanged and the reducer is a pass-through."""

    def __init__(self, module, bucket_bytes=None, process_group=None, broadcast_state=True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        if bucket_bytes is None:
            # the LAST bucket's all-reduce is the one backward cannot hide: at most 32 MiB, but at least four
            # buckets for a small model (15M parameters = 60 MB of gradients -> 15 MB buckets), never under 4 MiB
            total = 4 * sum(p.numel() for p in self.params)
            bucket_bytes = min(32 << 20, max(4 << 20, total // 4))
        params = list(reversed(self.params))              # backward order ~ reverse registration order
        self.buckets = []                                 # (flat, [params], [views])
        cur, cur_n = [], 0
        for p in params:
            cur.append(p)
            cur_n += p.numel() * 4
            if cur_n >= bucket_bytes:
                self._make_bucket(cur)
                cur, cur_n = [], 0
        if cur:
            self._make_bucket(cur)
        self._pending = []
        # gloo with GPU buckets (rehearsals of the multi-rank path on one GPU): gloo's own asynchronous CUDA
        # all-reduce hangs when two ranks share a device (seen on the MI355X test boxes with plain torch
        # tensors too), so that combination is exchanged synchronously through host memory in finish().
        # RCCL ("nccl") takes the asynchronous path.
        self._host_staged = (self.world > 1 and dist.get_backend(process_group) == "gloo"
                             and self.params and self.params[0].is_cuda)
        if self.world > 1 and broadcast_state:
            self.broadcast_state(module)
        self._hooks = []
        self._paused = False
        self._module = module
        # RCCL averages inside the collective (ReduceOp.AVG); gloo has no AVG, so sums are scaled in finish()
        self._avg_in_collective = self.world > 1 and dist.get_backend(process_group) == "nccl"
        self._reset()
        if self.world > 1:
            for bi, (_, ps, _) in enumerate(self.buckets):
                for p in ps:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def broadcast_state(self, module):
        """One-time copy of rank 0's parameters and buffers to every rank (what DDP's constructor does for the
        reference, init_utils.py:83-96): replicas start identical whatever each rank's seed was."""
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                if self._host_staged:
                    h = t.detach().cpu()
                    dist.broadcast(h, src=src, group=self.group)
                    t.copy_(h)
                else:
                    dist.broadcast(t.detach(), src=src, group=self.group)

    def _make_bucket(self, ps):
        n = sum(p.numel() for p in ps)
        flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
        views, off = [], 0
        for p in ps:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.buckets.append((flat, list(ps), views))

    def _reset(self):
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._next = 0                                    # buckets are exchanged strictly in index order

    def capturable_collectives(self):
        """True when the bucket all-reduces can be recorded into a HIP graph: RCCL ("nccl") on GPU buckets.  gloo's
        collectives run on the host and the host-staged rehearsal path synchronises: neither can be recorded."""
        return (self.world > 1 and not self._host_staged and bool(self.params) and self.params[0].is_cuda
                and dist.get_backend(self.group) == "nccl")

    def abort_step(self):
        """Forget a half-issued exchange (a graph capture that failed between hooks and finish())."""
        self._pending = []
        self._reset()

    def pause_hooks(self, paused):
        """While a HIP graph of forward + backward is being recorded the hooks must not fire (a collective is
        not part of the recorded step); reduce_grads() exchanges the gradients behind each replay instead."""
        self._paused = bool(paused)

    def reduce_grads(self, params, grads):
        """Exchange gradients that live outside ``p.grad`` (the static gradient tensors a replayed HIP graph writes
        into): gather them into the buckets, all-reduce bucket by bucket, and point ``p.grad`` at the averaged
        views for the optimizer.  Synchronous with respect to the stream (no overlap with backward: the replay is
        one launch)."""
        if self.world == 1:
            for p, g in zip(params, grads):
                p.grad = g
            return
        by_param = {id(p): g for p, g in zip(params, grads)}
        for p in self.params:
            p.grad = by_param.get(id(p))
        self._reset()
        self.finish()

    def _make_hook(self, bi):
        def hook(_p):
            if self._paused:
                return
            if self._launched[bi]:
                # a gradient arrived for a bucket that was already exchanged: a second backward without
                # finish() in between (finish() re-arms the reducer whichever zero_grad the caller uses)
                raise RuntimeError("GradBucketReducer: backward ran twice without finish(); call reducer.finish() "
                                   "after every backward (gradient accumulation over several backwards is not "
                                   "supported)")
            self._ready[bi] += 1
            # every rank issues the collectives in the same (index) order, whatever order its gradients
            # arrive in: a bucket goes out once it and all buckets before it are complete
            while self._next < len(self.buckets) and self._ready[self._next] == len(self.buckets[self._next][1]):
                self._launch(self._next)
                self._next += 1
        return hook

    def _launch(self, bi):
        flat, ps, views = self.buckets[bi]
        if flat.is_cuda:
            from . import ops
            ops.join_wgrad_lane()                         # the gradients about to be copied may still be on the side stream
        have = [(v, p.grad) for v, p in zip(views, ps) if p.grad is not None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(views, ps):
            if p.grad is None:
                v.zero_()                                 # parameter unused this step: contributes zero
            p.grad = v
        self._launched[bi] = True
        if self._host_staged:
            return                                        # exchanged in finish() through host memory
        op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        self._pending.append(dist.all_reduce(flat, op=op, group=self.group, async_op=True))

    def zero_grad(self):
        """Same effect as ``optimizer.zero_grad(set_to_none=True)`` / ``model.zero_grad()``: any of the three
        may be used (the reducer re-arms itself in finish(), not here)."""
        for p in self.params:
            p.grad = None

    def finish(self):
        """Wait for the outstanding all-reduces and turn sums into means."""
        if torch.cuda.is_available() and self.params and self.params[0].is_cuda:
            from . import ops
            ops.handoff_poll(self.params[0].device)       # collectives ran beside backward: did a hand-off launch starve?
        if self.world == 1:
            return
        for bi in range(self._next, len(self.buckets)):   # buckets holding parameters that got no gradient,
            self._launch(bi)                              # and everything queued behind them, in index order
        self._next = len(self.buckets)
        for w in self._pending:
            w.wait()
        self._pending = []
        if self._host_staged:
            for flat, _, _ in self.buckets:
                h = flat.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                flat.copy_(h)
        if not self._avg_in_collective:
            torch._foreach_mul_([flat for flat, _, _ in self.buckets], 1.0 / self.world)
        self._reset()                                     # armed for the next backward

    def sync_buffers(self):
        """Re-broadcast rank 0's buffers (BatchNorm running statistics of the lip encoder).  torch DDP, which the
        reference trains under (accelerate, broadcast_buffers=True), does this before every forward; here the
        training loop calls it where it matters: before evaluation and before saving a checkpoint (and, to
        reproduce DDP exactly, before each forward)."""
        if self.world == 1:
            return
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        with torch.no_grad():
            for t in self._module.buffers():
                if self._host_staged:
                    h = t.detach().cpu()
                    dist.broadcast(h, src=src, group=self.group)
                    t.copy_(h)
                else:
                    dist.broadcast(t.detach(), src=src, group=self.group)

    def remove(self):
        for h in self._hooks:
            h.remove()


class LocalGrads:
    """Stand-in for GradBucketReducer where the OPTIMIZER exchanges the gradients (ShardedClipAdamW: reduce-scatter
    -> update of the shard -> all-gather): backward leaves local gradients in ``p.grad`` and nothing is all-reduced.
    Same surface as the reducer for the training loop and for GraphedTrainStep (which then records forward +
    backward and runs the sharded step behind each replay)."""

    def __init__(self, module, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.buckets = []

    def capturable_collectives(self):
        return False

    def pause_hooks(self, paused):
        pass

    def abort_step(self):
        pass

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def finish(self):
        pass

    def reduce_grads(self, params, grads):
        for p, g in zip(params, grads):
            p.grad = g


class ShardedClipAdamW:
    """The optimizer step of the reference's loop (clip_grad_norm_ + AdamW, train.yaml:20-31,59, trainer.py:61-64)
    sharded over the data-parallel ranks:

        reduce-scatter(mean) of the flat gradient  ->  sum of squares of the local shard, all-reduce of ONE float
        ->  dtt_clip_adamw_ex on the shard (global clip coefficient, AdamW with this rank's 1/W of the moments)
        ->  all-gather of the updated flat parameters.

    Against all-reduce + a replicated update this moves the same bytes over xGMI (reduce-scatter + all-gather = one
    all-reduce) but touches 1/W of the parameter / moment bytes per GPU (206M parameters x (p, g, m, v) = 3.3 GB per
    step at cfg5) and keeps 1/W of the optimizer state.  Parameters are re-pointed at views of one flat fp32 buffer
    (padded to a multiple of 256 x world) so that the collectives run on whole buffers.

    ``weight_decay_of(name) -> float`` gives the per-parameter decay (the reference: 0 for names containing "bias",
    init_utils.py:63-80).  After step() ``p.grad`` still holds the LOCAL gradient; the averaged one exists only as
    this rank's shard.  Checkpointing the moments needs a gather of ``state_shard()`` over the ranks."""

    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, weight_decay_of=None,
                 max_norm=None, process_group=None, force_collectives=False):
        from .optim import ClipAdamW
        self._force = force_collectives       # issue the collectives even in a one-rank group (single-GPU RCCL probe)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        gran = 256 * self.world
        self.n_pad = (n + gran - 1) // gran * gran
        self.k = self.n_pad // self.world
        self.flat_p = torch.zeros(self.n_pad, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(self.n_pad, dtype=torch.float32, device=dev)
        self._host_staged = (self.world > 1 and dist.get_backend(process_group) == "gloo" and dev.type == "cuda")
        offs, o = [], 0
        with torch.no_grad():
            for p in self.params:
                self.flat_p[o:o + p.numel()].copy_(p.reshape(-1))
                p.data = self.flat_p[o:o + p.numel()].view_as(p)
                offs.append(o)
                o += p.numel()
        self.gviews = [self.flat_g[a:a + p.numel()].view_as(p) for a, p in zip(offs, self.params)]
        lo, hi = self.rank * self.k, (self.rank + 1) * self.k
        self.pshard = self.flat_p[lo:hi]
        self.gshard = torch.zeros(self.k, dtype=torch.float32, device=dev)
        groups = {}
        self.segments = []                                    # (name, leaf view of the parameter shard, offset in gshard)
        for (name, p), a in zip(named, offs):
            s0, s1 = max(a, lo), min(a + p.numel(), hi)
            if s0 >= s1:
                continue
            seg = self.flat_p[s0:s1].detach().requires_grad_(True)
            wd = weight_decay if weight_decay_of is None else weight_decay_of(name)
            groups.setdefault(float(wd), []).append(seg)
            self.segments.append((name, seg, s0 - lo, s1 - lo))
        if not self.segments:
            raise RuntimeError("ShardedClipAdamW: this rank's shard is empty (fewer parameters than 256 x world)")
        self.inner = ClipAdamW([{"params": ps, "weight_decay": wd} for wd, ps in groups.items()], lr=lr, betas=betas,
                               eps=eps, max_norm=max_norm)
        self.param_groups = self.inner.param_groups           # lr schedulers act on these

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def _reduce_scatter_mean(self):
        if self.world == 1 and not self._force:
            self.gshard.copy_(self.flat_g)
            return
        if self._host_staged or dist.get_backend(self.group) == "gloo":      # rehearsal path: gloo has no reduce-scatter
            h = self.flat_g.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            self.gshard.copy_(h[self.rank * self.k:(self.rank + 1) * self.k] / self.world)
            return
        dist.reduce_scatter_tensor(self.gshard, self.flat_g, op=dist.ReduceOp.AVG, group=self.group)

    def _all_reduce_sum(self, t):
        if self.world == 1 and not self._force:
            return t
        if self._host_staged:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _all_gather_params(self):
        if self.world == 1 and not self._force:
            return
        if self._host_staged or dist.get_backend(self.group) == "gloo":
            parts = [torch.empty(self.k) for _ in range(self.world)]
            dist.all_gather(parts, self.pshard.cpu(), group=self.group)
            self.flat_p.copy_(torch.cat(parts))
            return
        dist.all_gather_into_tensor(self.flat_p, self.pshard.clone(), group=self.group)

    @torch.no_grad()
    def step(self):
        """Returns the global (unclipped) gradient norm as a 1-element device tensor when max_norm is set."""
        if self.flat_p.is_cuda:
            from . import ops
            ops.join_wgrad_lane()
        have = [(v, p.grad) for v, p in zip(self.gviews, self.params) if p.grad is not None]
        missing = [v for v, p in zip(self.gviews, self.params) if p.grad is None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v in missing:
            v.zero_()
        self._reduce_scatter_mean()
        for _, seg, a, b in self.segments:
            seg.grad = self.gshard[a:b]
        ss = self._all_reduce_sum(self.inner.grad_sumsq())
        out = self.inner.step(sumsq=ss)
        self._all_gather_params()
        return out

    def state_shard(self):
        """{parameter name: (offset of the segment in the flat parameter, exp_avg, exp_avg_sq)} of this rank."""
        st = self.inner.state
        return {name: (a + self.rank * self.k, st[seg]["exp_avg"], st[seg]["exp_avg_sq"])
                for name, seg, a, _ in self.segments if seg in st}
