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
every step).

``capturable=True`` makes ``step()`` safe to record in a HIP graph (torch.cuda.graph): the step
count lives on the device and is incremented there, and the table of gradient pointers /
learning rates is uploaded from ONE pinned buffer, which ``refresh_hyper()`` rewrites before
each replay (after the scheduler has moved ``group["lr"]``)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import lib, check, stream_ptr, ptr, DttError


# DttOptTensor of include/dtt_hip.h (48 bytes)
_TAB = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("lr", "<f4"), ("wd", "<f4")])


class ClipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=None,
                 capturable=False, grad_scale=1.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("ClipAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_norm = max_norm
        self.capturable = capturable
        self.grad_scale = float(grad_scale)      # folded into the clip coefficient (1 / world after a SUM all-reduce)
        self._built = None
        b0 = self.param_groups[0]
        for g in self.param_groups:
            if tuple(g["betas"]) != tuple(b0["betas"]) or g["eps"] != b0["eps"]:
                raise ValueError("ClipAdamW: betas and eps must be the same in every group")

    def load_state_dict(self, state_dict):
        """The flat moment buffers and the step counter are rebuilt from the loaded per-parameter state on the
        next step (without this the loaded ``exp_avg`` / ``exp_avg_sq`` / ``step`` would be ignored once the
        optimizer has stepped: the flat buffers would keep the old moments)."""
        super().load_state_dict(state_dict)
        self._built = None
        # rebuild NOW: torch hands loaded tensors over without copying them when device and dtype already match,
        # so the loaded moments may alias the flat buffers of the optimizer that produced the state_dict
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if ps and all(p.is_cuda for p in ps):
            self._build()

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.setdefault("capturable", False)
        self.__dict__.setdefault("grad_scale", 1.0)
        self._built = None

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
                step = int(float(st["step"])) if step is None else step
            st["exp_avg"] = m[off:off + p.numel()].view_as(p)
            st["exp_avg_sq"] = v[off:off + p.numel()].view_as(p)
        self._step = step or 0
        self._step_t = torch.tensor(float(self._step))       # one tensor shared by every parameter's state
        self._step_dev = torch.full((1,), self._step, dtype=torch.int32, device=dev) if self.capturable else None
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
    def grad_sumsq(self):
        """Sum of squares of the gradients this optimizer holds, as a 1-element device tensor (deterministic
        two-stage reduction).  The sharded data-parallel step all-reduces it over the ranks and hands it back to
        ``step(sumsq=...)``, so that every shard clips by the GLOBAL norm."""
        if self._built is None:
            self._build()
        b = self._built
        self._upload([0 if p.grad is None else p.grad.data_ptr() for p in b["params"]])
        out = torch.zeros(1, dtype=torch.float32, device=b["m"].device)
        check(lib().dtt_grad_sumsq(ptr(b["dev_tab"]), len(b["params"]), ptr(b["chunks"]), b["chunks"].shape[0],
                                   ptr(b["partial"]), ptr(out), stream_ptr()), "dtt_grad_sumsq")
        return out

    @torch.no_grad()
    def step(self, closure=None, sumsq=None):
        """One clipped AdamW update.  Returns the (unclipped) total gradient norm as a 1-element device
        tensor when ``max_norm`` is set, else the closure's loss.  ``sumsq`` (1-element fp32 device tensor):
        externally merged squared norm (see grad_sumsq)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from . import ops
        ops.join_wgrad_lane()                                 # (a no-op after a complete backward pass: it joins itself)
        ops.handoff_poll()                                    # hand-off LSTM kernels of this backward (non-blocking)
        b = self._built
        if b is None or [p.data_ptr() for p in b["params"]] != b["ptrs"]:
            if torch.cuda.is_current_stream_capturing():
                # built inside a capture, the moment buffers would belong to the graph and be re-zeroed by every replay
                raise DttError("ClipAdamW: call prepare() before recording step() in a HIP graph")
            self._build()
            b = self._built
        grads = [p.grad for p in b["params"]]
        if not b["checked"]:                                  # layout contract, checked on the first step
            for gr in grads:
                if gr is not None and (gr.dtype != torch.float32 or not gr.is_contiguous() or gr.is_sparse
                                       or not gr.is_cuda):
                    raise DttError("ClipAdamW needs dense contiguous fp32 gradients on the GPU")
            b["checked"] = True
        gptrs = [0 if gr is None else gr.data_ptr() for gr in grads]
        if torch.cuda.is_current_stream_capturing():
            # recording: the gradient addresses are known now (graph memory), the upload itself stays outside the
            # graph -- refresh_hyper() sends the table ahead of every replay, stream-ordered, with the current lr
            for t in b["tabs"]:
                t["g"] = gptrs
        else:
            self._upload(gptrs)
        self._step += 1
        self._step_t.fill_(float(self._step))
        if self.capturable:
            self._step_dev.add_(1)                            # a device op: replayed with the graph
        g0 = self.param_groups[0]
        mx = float(self.max_norm) if self.max_norm else 0.0
        check(lib().dtt_clip_adamw_ex(ptr(b["dev_tab"]), len(b["params"]), ptr(b["chunks"]), b["chunks"].shape[0],
                                      ptr(b["partial"]), C.c_float(mx), C.c_float(g0["betas"][0]),
                                      C.c_float(g0["betas"][1]), C.c_float(g0["eps"]), self._step,
                                      ptr(self._step_dev), ptr(sumsq), C.c_float(self.grad_scale), ptr(b["norm"]),
                                      stream_ptr()), "dtt_clip_adamw_ex")
        return b["norm"] if mx > 0 else loss

    def prepare(self):
        """Allocate the flat moment buffers / tables now (must happen outside a graph capture)."""
        if self._built is None:
            self._build()

    def _upload(self, gptrs=None):
        """Table of {p, g, m, v, n, lr, wd} -> device, through one of two pinned copies (the copy issued two
        uploads ago has been read by the time its buffer is rewritten: one event wait, normally already past)."""
        b = self._built
        k = b["flip"]
        b["flip"] = 1 - k
        if b["events"][k] is not None:
            b["events"][k].synchronize()
        t = b["tabs"][k]
        if gptrs is not None:
            t["g"] = gptrs
        t["lr"] = np.array([g["lr"] for g in self.param_groups], dtype=np.float32)[b["group_of"]]
        t["wd"] = np.array([g["weight_decay"] for g in self.param_groups], dtype=np.float32)[b["group_of"]]
        b["dev_tab"].copy_(b["hosts"][k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        b["events"][k] = ev

    def refresh_hyper(self):
        """capturable mode, ahead of each graph replay (same stream): upload the table with the groups' current
        lr / weight decay and the gradient addresses noted while recording.  No host synchronisation with the
        previous replay is needed: the upload is stream-ordered between the two replays."""
        if not self.capturable or self._built is None:
            raise DttError("refresh_hyper() is for a capturable ClipAdamW whose step() has been recorded")
        self._upload()

    def note_gradients(self, grads):
        """capturable mode: the gradient tensors the recorded step() reads (one per parameter, None allowed).  A graph
        owner calls this after eager step() calls on the same optimizer, which noted their own gradients' addresses."""
        b = self._built
        if b is None:
            raise DttError("note_gradients() before the optimizer was built")
        gptrs = [0 if g is None else g.data_ptr() for g in grads]
        for t in b["tabs"]:
            t["g"] = gptrs

    def note_recorded_step(self):
        """Undo the host-side count of a step() that was only recorded into a graph (recording executes nothing)."""
        self._step -= 1
        self._step_t.fill_(float(self._step))

    def note_replayed_step(self):
        """Host-side bookkeeping for a step made by a graph replay (the device count moved by itself)."""
        self._step += 1
        self._step_t.fill_(float(self._step))
