"""Gradient clipping + AdamW in two launches (libdtt_hip: dtt_clip_adamw).

Replaces, for the training step that follows the hot path, the reference's
``clip_grad_norm_(model.parameters(), max_grad_norm)`` (base_trainer.py:478-490,
train.yaml:59) followed by ``torch.optim.AdamW.step()`` (train.yaml:20-23,
trainer.py:61-64).  Same arithmetic as ``torch.optim.AdamW`` (decoupled decay,
bias correction, eps outside the square root) and ``clip_grad_norm_`` (2-norm
over all gradients, coefficient ``min(1, max_norm / (norm + 1e-6))``); the
gradients themselves are left unscaled (the clipped values only exist inside
the update).  fp32 parameters and gradients on one GPU; learning-rate
schedulers work as with any ``torch.optim.Optimizer`` (``group["lr"]`` is read
every step)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import lib, check, stream_ptr, ptr, DttError


# DttOptTensor of include/dtt_hip.h (48 bytes)
_TAB = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("lr", "<f4"), ("wd", "<f4")])


class ClipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("ClipAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_norm = max_norm
        self._built = None
        b0 = self.param_groups[0]
        for g in self.param_groups:
            if tuple(g["betas"]) != tuple(b0["betas"]) or g["eps"] != b0["eps"]:
                raise ValueError("ClipAdamW: betas and eps must be the same in every group")

    # ------------------------------------------------------------------ #
    def _build(self):
        plist = [(p, gi) for gi, g in enumerate(self.param_groups) for p in g["params"] if p.requires_grad]
        if not plist:
            raise DttError("ClipAdamW: no parameters")
        dev = plist[0][0].device
        for p, _ in plist:
            if p.dtype != torch.float32 or not p.is_cuda or p.device != dev or not p.is_contiguous():
                raise DttError("ClipAdamW needs contiguous fp32 parameters on one GPU")
        # moments in two flat buffers, 16-byte aligned slices; the per-parameter views are the optimizer state
        offs, o = [], 0
        for p, _ in plist:
            offs.append(o)
            o += (p.numel() + 3) // 4 * 4
        m = torch.zeros(o, dtype=torch.float32, device=dev)
        v = torch.zeros(o, dtype=torch.float32, device=dev)
        step = None
        for (p, _), off in zip(plist, offs):
            st = self.state[p]
            if "exp_avg" in st:                               # loaded state: carry it over
                m[off:off + p.numel()].copy_(st["exp_avg"].reshape(-1))
                v[off:off + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                step = int(st["step"]) if step is None else step
            st["exp_avg"] = m[off:off + p.numel()].view_as(p)
            st["exp_avg_sq"] = v[off:off + p.numel()].view_as(p)
        self._step = step or 0
        self._step_t = torch.tensor(float(self._step))       # one tensor shared by every parameter's state
        for p, _ in plist:
            self.state[p]["step"] = self._step_t
        ch = lib().dtt_clip_adamw_chunk()
        chunks = [(i, c) for i, (p, _) in enumerate(plist) for c in range((p.numel() + ch - 1) // ch)]
        n = len(plist)
        hosts, tabs = [], []
        for _ in range(2):                                    # the table goes up by an async copy: two pinned copies
            h = torch.zeros(n * _TAB.itemsize, dtype=torch.uint8).pin_memory()
            t = h.numpy().view(_TAB)
            t["p"] = [p.data_ptr() for p, _ in plist]
            t["m"] = [m.data_ptr() + 4 * off for off in offs]
            t["v"] = [v.data_ptr() + 4 * off for off in offs]
            t["n"] = [p.numel() for p, _ in plist]
            hosts.append(h)
            tabs.append(t)
        self._built = dict(
            params=[p for p, _ in plist], group_of=np.array([gi for _, gi in plist]), m=m, v=v, hosts=hosts, tabs=tabs,
            events=[None, None], flip=0,
            dev_tab=torch.empty(n * _TAB.itemsize, dtype=torch.uint8, device=dev),
            chunks=torch.tensor(chunks, dtype=torch.int32, device=dev).contiguous(),
            partial=torch.empty(len(chunks), dtype=torch.float32, device=dev),
            norm=torch.zeros(1, dtype=torch.float32, device=dev),
            ptrs=[p.data_ptr() for p, _ in plist], checked=False)

    @torch.no_grad()
    def step(self, closure=None):
        """One clipped AdamW update.  Returns the (unclipped) total gradient norm as a 1-element device
        tensor when ``max_norm`` is set, else the closure's loss."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        b = self._built
        if b is None or [p.data_ptr() for p in b["params"]] != b["ptrs"]:
            self._build()
            b = self._built
        grads = [p.grad for p in b["params"]]
        if not b["checked"]:                                  # layout contract, checked on the first step
            for gr in grads:
                if gr is not None and (gr.dtype != torch.float32 or not gr.is_contiguous() or gr.is_sparse
                                       or not gr.is_cuda):
                    raise DttError("ClipAdamW needs dense contiguous fp32 gradients on the GPU")
            b["checked"] = True
        k = b["flip"]
        b["flip"] = 1 - k
        if b["events"][k] is not None:
            b["events"][k].synchronize()                      # the copy issued two steps ago has read this table
        t = b["tabs"][k]
        t["g"] = [0 if gr is None else gr.data_ptr() for gr in grads]
        t["lr"] = np.array([g["lr"] for g in self.param_groups], dtype=np.float32)[b["group_of"]]
        t["wd"] = np.array([g["weight_decay"] for g in self.param_groups], dtype=np.float32)[b["group_of"]]
        b["dev_tab"].copy_(b["hosts"][k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        b["events"][k] = ev
        self._step += 1
        self._step_t.fill_(float(self._step))
        g0 = self.param_groups[0]
        mx = float(self.max_norm) if self.max_norm else 0.0
        check(lib().dtt_clip_adamw(ptr(b["dev_tab"]), len(b["params"]), ptr(b["chunks"]), b["chunks"].shape[0],
                                   ptr(b["partial"]), C.c_float(mx), C.c_float(g0["betas"][0]), C.c_float(g0["betas"][1]),
                                   C.c_float(g0["eps"]), self._step, ptr(b["norm"]), stream_ptr()), "dtt_clip_adamw")
        return b["norm"] if mx > 0 else loss
