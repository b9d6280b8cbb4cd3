"""Data-parallel gradient exchange for one process per GPU (RCCL over xGMI).

The reference has no comm code of its own: HF accelerate wraps the model in
torch DDP (src/utils/init_utils.py:83-96, train.py:87,101).  This is a
from-scratch replacement sized for MI355X: gradients live in a few large flat
fp32 buckets (views, no copies) laid out in reverse-forward order (decoder ->
latent -> encoder, the order backward produces them); a bucket's all-reduce is
issued asynchronously the moment its last gradient has been accumulated, so
the collective overlaps the rest of backward.  xGMI is point-to-point, so few
large messages beat many small ones: the default bucket is 32 MiB.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, module, bucket_bytes=32 << 20, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()                                  # backward order ~ reverse registration order
        self.buckets = []                                 # (flat, [params])
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
        self._hooks = []
        for bi, (_, ps) in enumerate(self.buckets):
            for p in ps:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self._ready = [0] * len(self.buckets)

    def _make_bucket(self, ps):
        n = sum(p.numel() for p in ps)
        flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
        off = 0
        for p in ps:
            p.grad = flat[off:off + p.numel()].view_as(p)   # gradients accumulate straight into the bucket
            off += p.numel()
        self.buckets.append((flat, list(ps)))

    def _make_hook(self, bi):
        def hook(_p):
            self._ready[bi] += 1
            if self._ready[bi] == len(self.buckets[bi][1]):
                self._launch(bi)
        return hook

    def _launch(self, bi):
        if self.world > 1:
            flat = self.buckets[bi][0]
            self._pending.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def zero_grad(self):
        """Keep the bucket views (set_to_none would detach them)."""
        for flat, _ in self.buckets:
            flat.zero_()
        self._ready = [0] * len(self.buckets)

    def finish(self):
        """Wait for the outstanding all-reduces and turn sums into means."""
        if self.world > 1:
            for bi, r in enumerate(self._ready):          # buckets holding parameters that got no gradient
                if r != len(self.buckets[bi][1]):
                    self._launch(bi)
        for w in self._pending:
            w.wait()
        self._pending = []
        if self.world > 1:
            for flat, _ in self.buckets:
                flat.mul_(1.0 / self.world)

    def remove(self):
        for h in self._hooks:
            h.remove()
