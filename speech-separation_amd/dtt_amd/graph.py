"""The training step as ONE HIP graph.

A DTTNet step is a chain of ~2 400 short kernel launches; issued one by one from
Python through ctypes the host needs longer to enqueue them than the GPU needs
to run them (round-1 profile: 68 of 81 ms on the AV cross-attention model).
MI355X-first answer: record the whole step once -- STFT of mixture and sources,
model forward, PIT-L1 loss, backward, gradient clip + AdamW -- into a HIP graph
on torch's capture stream and replay it; the host then issues one launch per
step.  Everything the kernels touch is torch-owned memory from the graph's
private pool, the C ABI takes the stream as an argument, and no entry point
synchronises or allocates, so the capture needs nothing special from the
library.

What the reference's loop does per step (src/trainer/trainer.py:37-72) maps to:

    step = GraphedTrainStep(model, criterion, optimizer, example_batch)   # records
    for batch in loader:
        loss = step(batch)            # copies the batch into the static inputs, replays
        lr_scheduler.step()           # group["lr"] is re-read before the next replay

With more than one rank on RCCL the bucketed gradient all-reduces are recorded
INTO the graph (``overlap_comm``): the reducer's hooks fire during the capture,
each bucket's collective becomes a side branch of the graph that starts when the
bucket's last gradient is written and joins ahead of the optimizer step, so the
exchange overlaps the rest of backward exactly as in the eager path (and as
DDP's does for the reference, trainer.py:58) and the whole step is still one
launch.  Where collectives cannot be recorded (gloo rehearsals, or a capture
that fails) the recorded part ends after backward; the exchange
(GradBucketReducer.reduce_grads) and the optimizer step run behind the replay,
eagerly.
"""
from __future__ import annotations

import os

import torch

from . import ops
from ._lib import DttError


def _copy_into(dst, src):
    """Recursively copy a batch (tensors / lists / dicts of tensors) into its static twin."""
    if torch.is_tensor(dst):
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_into(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _copy_into(d, s)


def _clone_static(x):
    if torch.is_tensor(x):
        return x.clone()
    if isinstance(x, dict):
        return {k: _clone_static(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clone_static(v) for v in x]
    return x


class GraphedTrainStep:
    """``batch_fn(inputs) -> batch dict`` turns the static input tensors into the reference's batch dict
    (collate.py:72-83), e.g. by running the GPU STFT; it is recorded with the rest.  ``inputs`` is any
    nest of CUDA tensors with the shapes every later batch will have."""

    def __init__(self, model, criterion, optimizer, inputs, batch_fn, reducer=None, warmup=2, pack_plan=None,
                 handoff_kernels=True, overlap_comm=None):
        if not torch.cuda.is_available():
            raise DttError("GraphedTrainStep needs an MI355X")
        self.model, self.criterion, self.opt, self.reducer = model, criterion, optimizer, reducer
        self.world = reducer.world if reducer is not None else 1
        if overlap_comm is None:
            overlap_comm = os.environ.get("DTT_GRAPH_OVERLAP", "1") != "0"
        # collectives inside the graph: RCCL only (gloo's are host-side), and the optimizer must be recordable
        self.overlap = bool(overlap_comm and self.world > 1 and reducer is not None and reducer.capturable_collectives()
                            and getattr(optimizer, "capturable", False))
        if not getattr(optimizer, "capturable", False) and self.world == 1:
            raise DttError("GraphedTrainStep: build the ClipAdamW with capturable=True")
        self.static_in = _clone_static(inputs)
        self.batch_fn = batch_fn
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.graph = torch.cuda.CUDAGraph()
        self._replays = 0
        # derived weight tensors from one launch per step (ops._PackPlan), recorded by the first warm-up step.  On by
        # default since the table-driven kernel copies by vectors and tiles (0.26 ms instead of 0.6: cfg4 replay 39.55
        # -> 39.0 ms, same box, twice); pack_plan=False or DTT_PACK_PLAN=0 keeps the ~400 tiny launches (~2 us each
        # inside a graph)
        if pack_plan is None:
            pack_plan = os.environ.get("DTT_PACK_PLAN", "1") != "0"
        prev_plan = ops._PLAN.enabled
        ops.weight_plan(pack_plan)
        # handoff_kernels=False (or DTT_GRAPH_HANDOFF=0) keeps the cross-CU hand-off LSTM kernels (H >= 512) out of
        # the recorded step (streaming kernels instead).  They are safe to record since their counters are zeroed by
        # a kernel: as hipMemsetAsync NODES of the graph those resets made replays after a host-side stream sync
        # return wrong gradients (DESIGN finding 34).
        prev_handoff = ops._HANDOFF
        ops.handoff_kernels(handoff_kernels and os.environ.get("DTT_GRAPH_HANDOFF", "1") != "0")
        # The warm-up passes must leave the training state as they found it: parameters do not move (no optimizer
        # step), but every train-mode BatchNorm of the lip encoder would advance running_mean / running_var /
        # num_batches_tracked once per pass -- buffers are snapshotted and put back.  (With world > 1 the warm-up
        # does issue real collectives: every rank constructs its step at the same point.)
        saved = [(b, b.detach().clone()) for b in model.buffers()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # eager warm-up on the side stream (torch's recipe):
            for _ in range(max(1, warmup)):                # library plans, basis tables, MIOpen's find step
                self._eager_once(step_opt=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for b, v in saved:
                b.copy_(v)
        del saved
        if hasattr(optimizer, "prepare"):
            optimizer.prepare()
        ops.finalize_packs()
        for p in self.params:
            p.grad = None
        ops.reset_zero_arena()                             # zero-filled scratch must be born inside the capture
        if reducer is not None:
            reducer.pause_hooks(not self.overlap)
        try:
            try:
                self._record()
            except Exception as e:                          # noqa: BLE001
                if not self.overlap:
                    raise
                # the collectives could not be recorded on this stack: record forward + backward alone
                import sys
                print(f"[dtt_amd] recording the gradient all-reduces into the HIP graph failed ({type(e).__name__}: "
                      f"{e}); exchanging behind the replay instead", file=sys.stderr, flush=True)
                self.overlap = False
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                for p in self.params:
                    p.grad = None
                reducer.abort_step()
                reducer.pause_hooks(True)
                ops.reset_zero_arena()
                self._record()
        finally:
            ops.reset_zero_arena()                         # later eager work must not carve from the graph's arena
            ops.handoff_kernels(prev_handoff)              # only what is RECORDED avoids the hand-off kernels
            if reducer is not None:
                reducer.pause_hooks(False)
            if ops._PLAN.enabled != prev_plan:
                # the process-wide pack-plan switch goes back to what the caller had; the recorded step keeps
                # writing the plan's persistent buffers on every replay, so they stay alive with this object
                self._plan_keep = (ops._PLAN.outs, ops._PLAN.pairs, ops._PLAN.table)
                ops.weight_plan(prev_plan)
        self.graph_grads = [p.grad for p in self.params]   # written in place by every replay: keep them alive
        if self.world == 1 or self.overlap:
            self.opt.note_recorded_step()                  # recording executed nothing

    def _record(self):
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) may query events
            loss = self._fwd_bwd()
            if self.overlap:
                self.reducer.finish()                      # joins the collectives' branches (stream waits, recorded)
            if self.world == 1 or self.overlap:
                self.opt.step()
            self.static_loss = loss.detach()

    def _fwd_bwd(self):
        batch = self.batch_fn(self.static_in)
        batch.update(self.model(**batch))
        batch.update(self.criterion(**batch))
        batch["loss"].backward()
        ops.join_wgrad_lane()                              # (the backward pass joined its side stream already)
        ops.join_packs()
        return batch["loss"]

    def rearm(self):
        """After EAGER steps on the same model and optimizer (which left their own gradient tensors in ``.grad`` and
        their addresses in the optimizer's tables): point both back at the graph's static gradients."""
        for p, g in zip(self.params, self.graph_grads):
            p.grad = g
        if (self.world == 1 or self.overlap) and hasattr(self.opt, "note_gradients"):
            by_param = {id(p): g for p, g in zip(self.params, self.graph_grads)}
            self.opt.note_gradients([by_param.get(id(p)) for p in self.opt._built["params"]])

    def _eager_once(self, step_opt):
        for p in self.params:
            p.grad = None
        loss = self._fwd_bwd()
        if self.reducer is not None:
            self.reducer.finish()
        if step_opt:
            self.opt.step()
        return loss

    def __call__(self, inputs=None):
        """One training step.  Returns ``static_loss``: the SAME 0-dim device tensor every call, overwritten by the
        next replay -- ``float()`` / ``.clone()`` it to keep a value."""
        if inputs is not None:
            _copy_into(self.static_in, inputs)
        if self.world == 1 or self.overlap:
            self.opt.refresh_hyper()                       # table upload with the current lr, stream-ordered
            self.graph.replay()
            self.opt.note_replayed_step()
        else:
            self.graph.replay()
            self.reducer.reduce_grads(self.params, self.graph_grads)
            self.opt.step()
        self._replays += 1
        # the recorded step may hold hand-off LSTM launches: queue a (non-blocking) status read behind the replay;
        # a time-out raises DttError at the next call / loss / optimizer step, check() raises at once
        ops.handoff_poll(used=True)
        return self.static_loss

    def check(self):
        """Blocking: synchronise with the stream and raise DttError if a hand-off launch of any replay so far
        timed out (its step trained on garbage).  For loops that never read the loss on the host."""
        ops.handoff_check()
