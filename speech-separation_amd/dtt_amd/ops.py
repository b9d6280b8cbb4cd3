"""Autograd operators of the DTTNet path on top of libdtt_hip.so.

Every tensor that flows between these operators is channels-last
``[B, F, T, C]`` in the compute dtype (bf16 or fp32); parameters stay fp32 in
the reference's own layouts (state_dict contract) and are re-packed per call.
All heavy work is done by the HIP kernels; torch is used for allocation and
for (tiny) weight re-layouts only.
"""
from __future__ import annotations

import ctypes as C
import math

import os
import torch

from ._lib import lib, check, stream_ptr, ptr, dtype_code, gemm_raw, DttError

F32 = torch.float32


def _dt(t):
    return dtype_code(t.dtype)


def _empty(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


_ZARENA = {}


def _zeros_f32(shape, dev):
    """fp32 zeros carved out of a 64 MiB arena that is zero-filled by ONE kernel: the atomic weight-
    gradient targets and the identically-zero bias gradients need ~170 zeroed tensors per step, and a
    fill kernel each costs more than the arithmetic behind many of them.  A chunk is handed out once
    and never reused (a fresh arena replaces an exhausted one; the old one lives as long as its views)."""
    if isinstance(shape, int):
        shape = (shape,)
    n = 1
    for s in shape:
        n *= int(s)
    need = (n + 63) // 64 * 64                              # 256-byte granules keep every view 16-byte aligned
    key = (dev.type, dev.index)
    ar = _ZARENA.get(key)
    if ar is None or ar[1] + need > ar[0].numel():
        ar = [torch.zeros(max(16 << 20, need), dtype=F32, device=dev), 0]
        _ZARENA[key] = ar
    out = ar[0][ar[1]:ar[1] + n].view(shape)
    ar[1] += need
    return out


def reset_zero_arena():
    """Forget the current arena (its outstanding views stay valid).  Called around a HIP-graph capture: the
    zero fill has to be recorded with the step, and eager work afterwards must not carve from graph memory."""
    _ZARENA.clear()


# --------------------------------------------------------------------------- #
# Weight gradients on a second stream
# --------------------------------------------------------------------------- #
class _Lane:
    def __init__(self):
        self.stream, self.pending, self.keep, self.cb, self.fork = None, False, [], -1, None   # cb: task id with a join queued


_LANES = {}
_WGRAD_LANE = [os.environ.get("DTT_WGRAD_LANE", "1") == "1"]


def wgrad_lane(enabled=True):
    """Weight-gradient kernels (conv / Linear / LSTM weight and bias gradients: ~7 ms of a 40 ms cfg4 step) depend on
    nothing downstream but the optimizer, while the data-gradient chain they sit in is a string of launches that do not
    fill the chip at B = 16.  With the lane enabled each backward forks them onto a second stream (an event wait each
    way; under a HIP-graph capture: a parallel branch of the graph) and joins at the end of the backward pass
    (autograd's final callback), so they run beside the data-gradient kernels.  The tensors they read are kept alive
    until the join; their outputs are allocated on the launch stream before the fork.  Consumers of .grad that run
    INSIDE the backward pass (gradient accumulation into an existing .grad, post-accumulate hooks) must call
    join_wgrad_lane() first: GradBucketReducer's hooks do."""
    _WGRAD_LANE[0] = bool(enabled)


def _lane_ok(*params):
    """May the gradients of these forward inputs be produced on the side stream?  Only when nothing inside the
    backward pass reads them on the launch stream: every one a leaf (a slice or view of a parameter has a backward
    node of its own, which would read the gradient unsynchronised) whose .grad is still empty (AccumulateGrad then
    just keeps the tensor; with a .grad in place -- gradient accumulation over micro-batches, zero_grad(set_to_none=
    False) -- it adds on the launch stream).  Anything else computes on the launch stream."""
    for p in params:
        if p is None or not torch.is_tensor(p) or not p.requires_grad:
            continue
        if p.grad_fn is not None or p.grad is not None:
            return False
    return True


class _lane:
    """with _lane(dev, tensors to keep alive..., on=...): launches inside go to the device's side stream (on=False:
    they stay on the launch stream, see _lane_ok)."""

    def __init__(self, dev, *keep, on=True):
        self.dev, self.keep, self.ctx, self.on = dev, keep, None, on

    def __enter__(self):
        if not self.on or not _WGRAD_LANE[0] or self.dev.type != "cuda":
            return self
        ln = _LANES.setdefault(_dev_index(self.dev), _Lane())
        if ln.stream is None:
            ln.stream = torch.cuda.Stream(self.dev)
            ln.fork = torch.cuda.Event()
        # fork: one event, re-recorded per fork (a wait binds to the record that precedes it); wait_stream() would build
        # an Event object and resolve the current stream in Python each time: 50 us x 135 forks per step
        if torch.cuda.current_device() == ln.stream.device_index:
            ln.fork.record()
            ln.stream.wait_event(ln.fork)
        else:
            ln.stream.wait_stream(torch.cuda.current_stream(self.dev))
        ln.keep.extend(self.keep)
        ln.pending = True
        # joined when the backward pass that issued the work ends: one callback per pass, told apart by autograd's task
        # id (a flag alone would stay set after a pass that raised, and the next pass would end unjoined)
        task = torch._C._current_graph_task_id()
        if task >= 0 and ln.cb != task:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(join_wgrad_lane)
                ln.cb = task
            except RuntimeError:                   # not inside a backward pass: the caller joins
                pass
        self.ctx = torch.cuda.stream(ln.stream)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def join_wgrad_lane():
    """The launch stream waits for every weight-gradient launch issued so far (no host sync)."""
    for idx, ln in _LANES.items():
        if ln.pending:
            torch.cuda.current_stream(torch.device("cuda", idx)).wait_stream(ln.stream)
            ln.keep.clear()
            ln.pending = False


_VIDEO_LANE = [os.environ.get("DTT_VIDEO_LANE", "1") == "1"]
# the fork is recorded into HIP graphs too: a replay overlaps branches that fork at the graph's root (the lip encoder's
# forward: cfg4 replay 39.10 -> 38.32 ms), not those that fork mid-chain (DESIGN finding 43)
_VIDEO_IN_GRAPH = os.environ.get("DTT_VIDEO_LANE_GRAPH", "1") == "1"
_VIDEO_STREAMS = {}


def video_lane(enabled=True):
    """The lip encoder's forward depends on nothing but the clip, and its backward on nothing but the fusion's gradient:
    with the lane enabled the models run it on a stream of its own beside the audio encoder (autograd runs a node's
    backward on the stream of its forward and orders the streams where gradients cross).  Recorded into HIP graphs as
    well (DTT_VIDEO_LANE_GRAPH=0: one branch)."""
    _VIDEO_LANE[0] = bool(enabled)


class video_stream:
    """with video_stream(dev) as vs: y = lip_encoder(...)   then   y = vs.join(y) on the launch stream."""

    def __init__(self, dev):
        self.dev, self.st, self.ctx = dev, None, None

    def __enter__(self):
        if _VIDEO_LANE[0] and self.dev.type == "cuda" and (_VIDEO_IN_GRAPH or not torch.cuda.is_current_stream_capturing()):
            self.st = _VIDEO_STREAMS.get(_dev_index(self.dev))
            if self.st is None:
                self.st = _VIDEO_STREAMS[_dev_index(self.dev)] = torch.cuda.Stream(self.dev)
            self.st.wait_stream(torch.cuda.current_stream(self.dev))
            self.ctx = torch.cuda.stream(self.st)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        if self.st is not None:
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_stream(self.st)
            for t in tensors:
                t.record_stream(cur)                   # allocated on the side stream's pool, consumed on this one
        return tensors[0] if len(tensors) == 1 else tensors

    def defer(self, *tensors):
        """Join later: the loss wrappers call join_video_lane() before they read their inputs (tensors produced on the
        side stream that only the loss consumes, e.g. the sources' spectrograms)."""
        if self.st is not None:
            _VIDEO_DEFER.setdefault(_dev_index(self.dev), []).extend(tensors)


_VIDEO_DEFER = {}


def join_video_lane():
    """The launch stream waits for work deferred on the side stream (video_stream.defer); a no-op otherwise."""
    for idx, ts in _VIDEO_DEFER.items():
        if ts:
            cur = torch.cuda.current_stream(torch.device("cuda", idx))
            cur.wait_stream(_VIDEO_STREAMS[idx])
            for t in ts:
                t.record_stream(cur)
            ts.clear()


# --------------------------------------------------------------------------- #
# thin wrappers over the C entry points
# --------------------------------------------------------------------------- #
def conv_gemm(x, wp, bias, out, geom, res=None, phases=False, res00=None):
    """Implicit-GEMM 3x3 conv: x [B,Fi,Ti,Ci] -> out [B,Fo,To,Co]; wp [Co][9*Ci].  phases: force the four-parity
    form of a transposed conv (wins once a phase's K = Ci .. 4 Ci is long: the lip encoder's stride-2 convs)."""
    Co = wp.shape[0]
    if geom["ctrans"] and res is None and (Co <= 32 or phases):
        # measured: the four parity GEMMs win only while the tile is narrow (Co = 32: 499 vs 542 us at
        # full resolution); at Co >= 64 the K of a phase is too short and one 9-tap GEMM is faster
        return conv_gemm_transposed(x, wp, bias, out, geom, res00=res00)
    if res00 is not None:
        raise DttError("conv_gemm: a compact even-pixel residual needs the four-parity transposed form")
    M = geom["cB"] * geom["cFo"] * geom["cTo"]
    kw = dict(A=x, B=wp, C=out, M=M, N=Co, K=9 * geom["cCi"], a_ms=0, a_ks=1, b_ks=1, b_ns=wp.shape[1],
              c_ms=Co, c_ns=1, dtype=_dt(x), a_mode=1, **geom)
    if bias is not None:
        kw.update(bias=bias, bias_mode=1)
    if res is not None:
        kw.update(R=res, r_ms=Co, r_ns=1)
    gemm_raw(**kw)


def conv_gemm_transposed(x, wp, bias, out, geom, res00=None):
    """Transposed 3x3 stride-2 conv (ConvTranspose2d forward / data gradient of the stride-2 conv) as
    four GEMMs, one per output parity (fo%2, to%2): each touches only the taps that can reach such
    a pixel (1, 2, 2 and 4 of the 9), so no structural zeros are multiplied.  wp [Co][9*Ci].
    res00 [B, ceil(Fo/2), ceil(To/2), Co] (compact): added to the even-even output pixels -- the data gradient of a
    1x1 stride-2 conv over the same input (the ResNet downsample branch) lands exactly there."""
    Co, Ci = wp.shape[0], geom["cCi"]
    w4 = wp.view(Co, 3, 3, Ci)
    for pf in (0, 1):
        for pt in (0, 1):
            # taps {0, 2} or {1} per axis, as slices: indexing with a Python list would build an index tensor
            # on the host and copy it over synchronously -- a hidden stream sync in the middle of backward
            # (and, with a collective in flight on another stream, a deadlock between ranks)
            khs = slice(0, 3, 2) if pf else slice(1, 2)
            kws = slice(0, 3, 2) if pt else slice(1, 2)
            Fp, Tp = (geom["cFo"] - pf + 1) // 2, (geom["cTo"] - pt + 1) // 2
            if Fp <= 0 or Tp <= 0:
                continue
            wq = w4[:, khs, kws].reshape(Co, -1).contiguous()
            kw = dict(A=x, B=wq, C=out, M=geom["cB"] * Fp * Tp, N=Co, K=wq.shape[1], a_ms=0, a_ks=1, b_ks=1,
                      b_ns=wq.shape[1], c_ms=Co, c_ns=1, dtype=_dt(x), a_mode=1, cphase=1 + 2 * pf + pt, **geom)
            if bias is not None:
                kw.update(bias=bias, bias_mode=1)
            if res00 is not None and pf == 0 and pt == 0:
                if tuple(res00.shape) != (geom["cB"], Fp, Tp, Co) or not res00.is_contiguous():
                    raise DttError("conv_gemm_transposed: res00 must be a contiguous [B, ceil(Fo/2), ceil(To/2), Co] tensor")
                kw.update(R=res00, r_ms=Co, r_ns=1)
            gemm_raw(**kw)


def conv_wgrad(p_side, q_side, geom, M, N, on=True):
    """dWp[m][tap][n] = sum_p P[p][m] * Q[src(p,tap)][n]; geom: cFo/cTo = P grid, cFi/cTi = Q grid."""
    P = geom["cB"] * geom["cFo"] * geom["cTo"]
    dwp = _zeros_f32((M, 9, N,), p_side.device)
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * 9
    splitk = int(max(1, min(512, round(768 / tiles), (P + 255) // 256)))     # ~3 workgroups per CU: atomic tail
    with _lane(p_side.device, p_side, q_side, on=on):
        gemm_raw(A=p_side, B=q_side, C=dwp, M=M, N=N, K=P, batch=9, splitk=splitk, a_ms=1, a_ks=M, a_bs=0,
                 b_ks=N, b_ns=1, b_bs=0, c_ms=9 * N, c_ns=1, c_bs=N, dtype=_dt(p_side), atomic=1, out_f32=1,
                 b_mode=1, **geom)
    return dwp


def _splitk(M, N, K, batch=1):
    """Split-K factor for an atomically accumulated weight-gradient GEMM: enough workgroups
    (~2-3 per CU) to fill 256 CUs, no more -- every split adds M*N fp32 atomics."""
    tiles = -(-M // 128) * -(-N // 128) * batch
    return int(max(1, min(128, K // 256, round(512 / tiles))))


def colsum(x2d, rows, width, ld=None):
    """Deterministic column sums of a [rows][width] matrix -> fp32 [width]."""
    ld = width if ld is None else ld
    ept = 8 if x2d.dtype == torch.bfloat16 else 4
    if width < ept and ept % width == 0 and ld == width and rows % (ept // width) == 0 and rows >= 4096:
        # a few columns only (the 2S mask channels): fold rows so that a 16-byte chunk is one "row" and the
        # vector kernel applies (the scalar kernel would run 4 lanes of 256), then add the folded copies
        k = ept // width
        return colsum(x2d, rows // k, ept).view(k, width).sum(0)
    nch = int(max(1, min(1024, rows // 256)))
    part = _empty((nch, width), F32, x2d.device)
    check(lib().dtt_colsum(ptr(x2d), ptr(part), C.c_int64(rows), width, C.c_int64(ld), nch, _dt(x2d), stream_ptr()),
          "dtt_colsum")
    if nch == 1:
        return part[0]
    out = _empty((1, width), F32, x2d.device)
    check(lib().dtt_colsum(ptr(part), ptr(out), C.c_int64(nch), width, C.c_int64(width), 1, 0, stream_ptr()),
          "dtt_colsum")
    return out[0]


def _in_chunk(B, P):
    """Pixels per workgroup of the InstanceNorm reduction passes: ~2048 workgroups, chunks of >= 64 pixels, but
    down to 16 pixels (one sweep at 512 channels) while that leaves fewer than 1024 workgroups."""
    cap = max(1, 2048 // B)
    nch = max(1, min(P // 64 if P >= 64 else 1, cap))
    if B * nch < 1024:
        nch = max(nch, min(max(1, P // 16), cap))
    return (P + nch - 1) // nch


def in_stats(y, eps=1e-5):
    B, Fy, Ty, Cc = y.shape
    P = Fy * Ty
    chunk = _in_chunk(B, P)
    nch = (P + chunk - 1) // chunk
    part = _empty((B, nch, Cc, 2), F32, y.device)
    stats = _empty((B, Cc, 2), F32, y.device)
    check(lib().dtt_in_stats(ptr(y), ptr(part), ptr(stats), B, P, Cc, chunk, C.c_float(eps), _dt(y), stream_ptr()),
          "dtt_in_stats")
    return stats


def in_gelu_fwd(y, stats, Fz, Tz):
    B, Fy, Ty, Cc = y.shape
    z = _empty((B, Fz, Tz, Cc), y.dtype, y.device)
    check(lib().dtt_in_gelu_fwd(ptr(y), ptr(stats), ptr(z), B, Fy, Ty, Fz, Tz, Cc, _dt(y), stream_ptr()),
          "dtt_in_gelu_fwd")
    return z


def in_gelu_bwd(y, dz, stats):
    B, Fy, Ty, Cc = y.shape
    Fz, Tz = dz.shape[1], dz.shape[2]
    P = Fy * Ty
    chunk = _in_chunk(B, P)
    nch = (P + chunk - 1) // chunk
    part = _empty((B, nch, Cc, 2), F32, y.device)
    sums = _empty((B, Cc, 2), F32, y.device)
    dy = torch.empty_like(y)
    check(lib().dtt_in_gelu_bwd(ptr(y), ptr(dz), ptr(stats), ptr(part), ptr(sums), ptr(dy), B, Fy, Ty, Fz, Tz, Cc,
                                chunk, _dt(y), stream_ptr()), "dtt_in_gelu_bwd")
    return dy


def _fast3(x, Co):
    """The dedicated 3x3 stride-1 kernels need 32-multiples of output channels."""
    return Co % 32 == 0 and x.shape[3] % (8 if x.dtype == torch.bfloat16 else 4) == 0


def conv3x3_s1(x, wp, bias, res=None, want_stats=False, eps=1e-5):
    """x [B,F,T,Ci], wp [Co][9*Ci] -> y [B,F,T,Co] (+ InstanceNorm stats [B,Co,2])."""
    B, Fq, T, Ci = x.shape
    Co = wp.shape[0]
    y = _empty((B, Fq, T, Co), x.dtype, x.device)
    part = stats = None
    if want_stats:
        part = _empty((B, 4 * ((Fq + 15) // 16) * ((T + 15) // 16) + 4, Co, 3), F32, x.device)
        stats = _empty((B, Co, 2), F32, x.device)
    check(lib().dtt_conv3x3_s1(ptr(x), ptr(wp), ptr(bias), ptr(y), ptr(res), ptr(part), ptr(stats), B, Fq, T, Ci, Co,
                               C.c_float(eps), _dt(x), stream_ptr()), "dtt_conv3x3_s1")
    return y, stats


def binary(op, a, b):
    out = torch.empty_like(a)
    check(lib().dtt_binary(op, ptr(a), ptr(b), ptr(out), C.c_int64(a.numel()), _dt(a), stream_ptr()), "dtt_binary")
    return out


# --------------------------------------------------------------------------- #
# Gradient hand-over between the two consumers of one tensor
# --------------------------------------------------------------------------- #
class GradSlot:
    """A tensor with two consumers (the input of a TFC-TDF block: conv1[0] and skip_conv, block_tfc_tdf.py:40-41; an
    encoder skip: the down-sampling conv and the decoder's multiply, encoder.py:43-46 / decoder.py:43; the input of a
    ResNet block: conv1 and the residual / downsample branch, resnet.py:57-81) gets its gradient as the SUM of two
    partial gradients, which the autograd engine forms with a separate add kernel per tensor (29 per cfg4 step,
    848 MB written).  Both consumers are given one GradSlot: the one whose backward runs FIRST parks its partial
    gradient here and reports None; the one that runs second -- always a convolution's data gradient -- takes it as the
    residual input of its dgrad kernel (added in the epilogue, no extra pass) and reports the sum.  Which runs first is
    fixed by data dependence in every pairing used here (the parking node is upstream in backward order), except
    the ResNet downsample pairing, which handles both orders (Conv1x1s2.backward).
    A slot serves ONE backward pass of ONE forward; an unconsumed slot just means the plain autograd add happened."""
    __slots__ = ("g", "kind")

    def __init__(self):
        self.g, self.kind = None, None

    def put(self, g, kind="full"):
        self.g, self.kind = g, kind

    def take(self):
        g, k = self.g, self.kind
        self.g, self.kind = None, None
        return g, k


_GRAD_SLOTS = os.environ.get("DTT_GRAD_SLOT", "1") != "0"      # 0: plain autograd accumulation (A/B switch)


def new_slot(t):
    """A GradSlot for tensor `t` and its two consumers, or None when t needs no gradient (or DTT_GRAD_SLOT=0)."""
    return GradSlot() if (_GRAD_SLOTS and t.requires_grad and torch.is_grad_enabled()) else None


def _slot_take(slot, want="full"):
    if slot is None:
        return None
    g, k = slot.take()
    if g is not None and k != want:
        raise DttError(f"GradSlot holds a {k!r} partial gradient where {want!r} was expected")
    return g


# --------------------------------------------------------------------------- #
# conv3x3 (+ InstanceNorm + GELU)
# --------------------------------------------------------------------------- #
def _conv_geoms(kind, B, Fi, Ti, Ci, Co):
    """kind: 's1' same conv, 'down' stride-2 conv, 'up' transposed conv (stride 2, pad 1, out pad 1)."""
    if kind == "s1":
        Fo, To = Fi, Ti
        fwd = dict(cB=B, cFi=Fi, cTi=Ti, cCi=Ci, cFo=Fo, cTo=To, cstride=1, ctrans=0)
        dgrad = dict(cB=B, cFi=Fo, cTi=To, cCi=Co, cFo=Fi, cTo=Ti, cstride=1, ctrans=0)
        wg = dict(cB=B, cFi=Fi, cTi=Ti, cCi=Ci, cFo=Fo, cTo=To, cstride=1, ctrans=0)
    elif kind == "down":
        Fo, To = (Fi + 1) // 2, (Ti + 1) // 2
        fwd = dict(cB=B, cFi=Fi, cTi=Ti, cCi=Ci, cFo=Fo, cTo=To, cstride=2, ctrans=0)
        dgrad = dict(cB=B, cFi=Fo, cTi=To, cCi=Co, cFo=Fi, cTo=Ti, cstride=2, ctrans=1)
        wg = dict(cB=B, cFi=Fi, cTi=Ti, cCi=Ci, cFo=Fo, cTo=To, cstride=2, ctrans=0)
    elif kind == "up":
        Fo, To = 2 * Fi, 2 * Ti
        fwd = dict(cB=B, cFi=Fi, cTi=Ti, cCi=Ci, cFo=Fo, cTo=To, cstride=2, ctrans=1)
        dgrad = dict(cB=B, cFi=Fo, cTi=To, cCi=Co, cFo=Fi, cTo=Ti, cstride=2, ctrans=0)
        # wgrad: P side = x grid (Fi,Ti), Q side = dy grid (Fo,To) gathered with stride 2
        wg = dict(cB=B, cFi=Fo, cTi=To, cCi=Co, cFo=Fi, cTo=Ti, cstride=2, ctrans=0)
    else:
        raise DttError(f"unknown conv kind {kind}")
    return Fo, To, fwd, dgrad, wg


def _cast_perm(w, perm, dt):
    """Permute + dtype cast of a parameter in ONE copy kernel (permute().reshape().to() takes two)."""
    v = w.permute(*perm)
    out = torch.empty(v.shape, dtype=dt, device=w.device)
    out.copy_(v)
    return out


# --------------------------------------------------------------------------- #
# Pack plan: every derived weight tensor of a step from ONE launch
# --------------------------------------------------------------------------- #
class DttPackDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("sstride", C.c_int64 * 4), ("dstride", C.c_int64 * 4),
                ("numel", C.c_int64), ("shape", C.c_int32 * 4), ("src_dtype", C.c_int32), ("dst_dtype", C.c_int32)]


def _merge_dims(shape, ss, ds):
    """Drop size-1 dims and merge neighbours that are contiguous in BOTH views."""
    dims = [(n, a, b) for n, a, b in zip(shape, ss, ds) if n != 1]
    out = []
    for n, a, b in dims:
        if out and out[-1][1] == a * n and out[-1][2] == b * n:
            out[-1] = (out[-1][0] * n, a, b)
        else:
            out.append((n, a, b))
    return out or [(1, 1, 1)]


class _StridedSource:
    """A source view torch cannot express (negative strides: the data-gradient conv packs walk the 3 x 3 taps
    backwards), for a pack-plan pair: element strides and offset relative to `base`, and `now()` = the same values as
    a real tensor (the recording step's copy)."""

    def __init__(self, base, shape, strides, offset, now):
        self.base, self.shape, self._strides, self.offset, self.now = base, tuple(shape), tuple(strides), offset, now
        self.dtype = base.dtype

    def stride(self):
        return self._strides

    def numel(self):
        n = 1
        for k in self.shape:
            n *= k
        return n

    def data_ptr(self):
        return self.base.data_ptr() + self.offset * self.base.element_size()


class _PackPlan:
    """A training step derives ~400 small tensors from its parameters (bf16 casts, conv weights in tap-major
    order, stacked / transposed LSTM weights, padded TDF weights, ...): one 5 us launch each, strung between
    the big kernels of the step (2.3 ms of a 47 ms step, in a HIP graph too).  Nearly all of them are strided
    copies with a cast.  With the plan enabled, the first step records each as descriptors (source view ->
    view of a PERSISTENT destination buffer) and `prepare` (called by the models' forward) refreshes all the
    buffers with one dtt_multi_pack launch (pack.hip); the few recipes that are not plain copies (bias sums)
    run as recorded closures right behind it.  Recipes read the parameters as they are when `prepare` runs:
    parameters must not change between the start of forward and the end of backward, nor move in memory while
    the plan is enabled (training loops satisfy this; weight_plan() forgets the plan).  Off by default for plain eager
    use; GraphedTrainStep and bench.py's eager mode switch it on.  The table-driven kernel first walked every copy
    element by element (0.58 ms per cfg4 step: 4-byte reads at a stride for the transposed and tap-major copies, and
    no gain under graph replay: 47.3 ms with the plan, 46.5 without); by vectors and LDS tiles it takes 0.26 ms (cfg4
    replay 39.55 -> 39.0 ms, eager issue 36.3 -> 35.5).
    (A first form ran the recipes on a side stream joined in stages: as a branch of the step's HIP graph it
    made the replay 0.9 ms SLOWER than the chain of tiny launches it removed.)"""

    def __init__(self):
        self.enabled = False
        self.outs = {}          # key -> persistent packed tensor
        self.pairs = []         # (dst view, src view) of every recorded copy
        self.closures = {}      # key -> recipe that is not a plain copy
        self.fresh = {}         # key -> this step's result of a closure
        self.table = None       # (descs on device, chunks on device, nchunks), rebuilt when pairs grow
        self.ntab = 0

    def reset(self, enabled):
        self.__init__()
        self.enabled = enabled

    def build(self):
        descs, chunks = [], []
        for d, v in self.pairs:
            dims = _merge_dims(v.shape, v.stride(), d.stride())
            dims = [(1, 0, 0)] * (4 - len(dims)) + dims
            e = DttPackDesc()
            e.src, e.dst = v.data_ptr(), d.data_ptr()
            for i, (n, a, b) in enumerate(dims):
                e.shape[i], e.sstride[i], e.dstride[i] = n, a, b
            e.numel = v.numel()
            e.src_dtype, e.dst_dtype = dtype_code(v.dtype), dtype_code(d.dtype)
            for c in range(lib().dtt_multi_pack_nchunks(C.byref(e))):
                chunks += [len(descs), c]
            descs.append(e)
        arr = (DttPackDesc * len(descs))(*descs)
        dev = self.pairs[0][0].device
        dt_ = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        ct_ = torch.tensor(chunks, dtype=torch.int32).to(dev)
        self.table, self.ntab = (dt_, ct_, len(chunks) // 2), len(self.pairs)


_PLAN = _PackPlan()


def weight_plan(enabled=True):
    """Enable / disable (and forget) the pack plan."""
    _PLAN.reset(bool(enabled))


def prepare_packs():
    """Refresh every recorded packed tensor (start of a training step; a no-op when the plan is off or empty).
    Called by the models' forward."""
    pl = _PLAN
    if not pl.enabled:
        return
    if pl.pairs:
        if pl.table is None or pl.ntab != len(pl.pairs):
            if torch.cuda.is_current_stream_capturing():
                raise DttError("pack plan: new packed tensors were recorded after the table was built; run one eager "
                               "step and ops.finalize_packs() before capturing")
            pl.build()
        descs, chunks, n = pl.table
        check(lib().dtt_multi_pack(ptr(descs), ptr(chunks), n, stream_ptr()), "dtt_multi_pack")
    with torch.no_grad():
        pl.fresh = {k: fn() for k, fn in pl.closures.items()}


def finalize_packs():
    """Build the descriptor table of everything recorded so far (host -> device copies: call it outside a stream
    capture, as GraphedTrainStep does after its warm-up steps)."""
    pl = _PLAN
    if pl.enabled and pl.pairs and (pl.table is None or pl.ntab != len(pl.pairs)):
        pl.build()


def join_packs():
    """(Kept for callers of the first, side-stream form of the plan: nothing to join any more.)"""
    return None


def packed(key, recipe, spec=None):
    """The tensor `recipe()` derives from parameters.  With the plan enabled and `spec` = (shape, dtype,
    pairs(out) -> [(view of out, source view of the same shape), ...]) it is a persistent buffer refreshed by
    prepare_packs(); without a spec the recipe is re-run by prepare_packs()."""
    pl = _PLAN
    if not pl.enabled:
        return recipe()
    out = pl.outs.get(key)
    if out is not None:
        return out
    if key in pl.fresh:
        return pl.fresh[key]
    if spec is not None:
        shape, dt, pairs = spec
        with torch.no_grad():
            # destination first (zeros: padding stays zero for good), then the copies of the recording step
            out = torch.zeros(shape, dtype=dt, device=torch.device("cuda", torch.cuda.current_device()))
            pr = pairs(out)
            if all(d.shape == v.shape and len(_merge_dims(v.shape, v.stride(), d.stride())) <= 4 for d, v in pr):
                for d, v in pr:
                    d.copy_(v.now() if isinstance(v, _StridedSource) else v)
                pl.pairs += pr
                pl.outs[key] = out
                return out
    v = recipe()
    pl.closures[key] = recipe
    pl.fresh[key] = v
    return v


def _pp(tag, w, perm, dt):
    """_cast_perm(w, perm, dt) through the plan."""
    shape = tuple(w.shape[i] for i in perm)
    return packed(_wkey(tag, dt, w), lambda: _cast_perm(w, perm, dt), (shape, dt, lambda out: [(out, w.permute(*perm))]))


def _wkey(tag, dt, *ws):
    return (tag, dt) + tuple(w.data_ptr() for w in ws)


def _pack_fwd(w, kind, dt):
    if kind == "up":      # ConvTranspose2d weight [Ci, Co, 3, 3]
        return _pp("pfup", w, (1, 2, 3, 0), dt).view(w.shape[1], -1)
    return _pp("pf", w, (0, 2, 3, 1), dt).view(w.shape[0], -1)


def _pack_dgrad_s1_now(w, dt):
    if w.dtype == F32 and w.is_contiguous() and w.shape[2:] == (3, 3):
        out = torch.empty((w.shape[1], 9 * w.shape[0]), dtype=dt, device=w.device)
        check(lib().dtt_pack_conv3_dgrad(ptr(w), ptr(out), w.shape[0], w.shape[1], 1 if dt == torch.bfloat16 else 0,
                                         stream_ptr()), "dtt_pack_conv3_dgrad")
        return out
    return _cast_perm(w.flip(2, 3), (1, 2, 3, 0), dt).view(w.shape[1], -1)


def _pack_dgrad(w, kind, dt):
    if kind == "s1":      # out[ci][kh][kw][co] = w[co][ci][2-kh][2-kw]: taps rotated, channels swapped
        Co, Ci = w.shape[0], w.shape[1]

        def pairs(out):
            o4 = out.view(Ci, 3, 3, Co)
            if w.is_contiguous():          # one descriptor: the taps as a dim of stride -1 from the last tap
                src = _StridedSource(w, (Ci, 3, 3, Co), (9, -3, -1, 9 * Ci), 8,
                                     lambda: w.flip(2, 3).permute(1, 2, 3, 0))
                return [(o4, src)]
            return [(o4[:, kh, kw, :], w[:, :, 2 - kh, 2 - kw].t()) for kh in range(3) for kw in range(3)]
        return packed(_wkey("pds1", dt, w), lambda: _pack_dgrad_s1_now(w, dt), ((Ci, 9 * Co), dt, pairs))
    if kind == "down":
        return _pp("pddown", w, (1, 2, 3, 0), dt).view(w.shape[1], -1)
    return _pp("pdup", w, (0, 2, 3, 1), dt).view(w.shape[0], -1)                    # up: [Ci][tap][Co]


def _conv_backward(kind, x, w, dy, need_dx, tile=True, phases=False, res=None, res00=None):
    """dy: gradient of the raw conv output [B,Fo,To,Co]. Returns dx, dw (param layout, fp32).  tile=False keeps a
    stride-1 conv off the 16x16-pixel tile kernels (small images: Conv3x3).  res (shape of x) / res00 (compact,
    even pixels: see conv_gemm_transposed): a partial input gradient from x's other consumer, added in the data
    gradient's epilogue (GradSlot)."""
    B, Fi, Ti, Ci = x.shape
    Co = dy.shape[3]
    on = _lane_ok(w)                               # the weight gradient may leave the launch stream
    _, _, _, gd, gw = _conv_geoms(kind, B, Fi, Ti, Ci, Co)
    if kind == "s1" and tile and _fast3(dy, Ci) and _fast3(x, Co):
        if x.dtype == torch.bfloat16:
            need = lib().dtt_conv3x3_wgrad_ws_bytes(B, Fi, Ti, Ci, Co)
            if need > 0:        # partial sums through a workspace + one sum launch instead of fp32 atomics;
                dw = _empty((Co, Ci, 3, 3), F32, x.device)      # the sum writes the parameter's own layout
                with _lane(x.device, dy, x, on=on):
                    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
                    check(lib().dtt_conv3x3_wgrad_ws_param(ptr(dy), ptr(x), ptr(dw), ptr(ws), C.c_int64(need), B, Fi, Ti,
                                                           Ci, Co, 1, stream_ptr()), "dtt_conv3x3_wgrad_ws_param")
                dx = conv3x3_s1(dy, _pack_dgrad(w, kind, x.dtype), None, res=res)[0] if need_dx else None
                return dx, dw
            else:
                dwp = _zeros_f32((Co, 9, Ci,), x.device)
                with _lane(x.device, dy, x, on=on):
                    check(lib().dtt_conv3x3_wgrad(ptr(dy), ptr(x), ptr(dwp), B, Fi, Ti, Ci, Co, 1, stream_ptr()),
                          "dtt_conv3x3_wgrad")
        else:
            dwp = conv_wgrad(dy, x, gw, Co, Ci, on=on)
        with _lane(x.device, dwp, on=on):
            dw = dwp.view(Co, 3, 3, Ci).permute(0, 3, 1, 2).contiguous()
        dx = conv3x3_s1(dy, _pack_dgrad(w, kind, x.dtype), None, res=res)[0] if need_dx else None
        return dx, dw
    if kind == "up":
        dwp = conv_wgrad(x, dy, gw, Ci, Co, on=on)                      # [Ci][9][Co]
        dw = dwp.view(Ci, 3, 3, Co).permute(0, 3, 1, 2)
    else:
        dwp = conv_wgrad(dy, x, gw, Co, Ci, on=on)                      # [Co][9][Ci]
        dw = dwp.view(Co, 3, 3, Ci).permute(0, 3, 1, 2)
    with _lane(x.device, dwp, on=on):
        dw = dw.contiguous()
    dx = None
    if need_dx:
        dx = torch.empty_like(x)
        conv_gemm(dy, _pack_dgrad(w, kind, x.dtype), None, dx, gd, phases=phases, res=res, res00=res00)
    return dx, dw


class ConvNormGelu(torch.autograd.Function):
    """conv3x3 -> InstanceNorm2d(affine=False, eps 1e-5) -> exact GELU (-> crop).
    Reference: block_tfc_tdf.py:12-17, block_sampling.py:8-12,23-28, decoder.py:46-52."""

    @staticmethod
    def forward(ctx, x, w, b, kind, crop, slot=None):
        B, Fi, Ti, Ci = x.shape
        Co = w.shape[1] if kind == "up" else w.shape[0]
        Fo, To, gf, _, _ = _conv_geoms(kind, B, Fi, Ti, Ci, Co)
        ctx.slot = slot
        if kind == "s1" and _fast3(x, Co):
            y, stats = conv3x3_s1(x, _pack_fwd(w, kind, x.dtype), b, want_stats=True)
        else:
            y = _empty((B, Fo, To, Co), x.dtype, x.device)
            conv_gemm(x, _pack_fwd(w, kind, x.dtype), b, y, gf)
            stats = in_stats(y)
        Fz, Tz = (Fo, To) if crop is None else crop
        z = in_gelu_fwd(y, stats, Fz, Tz)
        ctx.save_for_backward(x, w, y, stats)
        ctx.kind = kind
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w, y, stats = ctx.saved_tensors
        dy = in_gelu_bwd(y, dz.contiguous(), stats)
        res = _slot_take(ctx.slot) if ctx.needs_input_grad[0] else None     # x's other consumer ran first (GradSlot)
        dx, dw = _conv_backward(ctx.kind, x, w, dy, ctx.needs_input_grad[0], res=res)
        # the bias feeds an affine-free InstanceNorm: its gradient is identically zero
        db = _zeros_f32((w.shape[1] if ctx.kind == "up" else w.shape[0],), x.device)
        return dx, dw, db, None, None, None


class ConvAdd(torch.autograd.Function):
    """skip_conv(x) + h  (block_tfc_tdf.py:41,46): conv3x3 with bias and a fused residual add."""

    @staticmethod
    def forward(ctx, x, w, b, h, slot=None):
        B, Fi, Ti, Ci = x.shape
        Co = w.shape[0]
        _, _, gf, _, _ = _conv_geoms("s1", B, Fi, Ti, Ci, Co)
        ctx.slot = slot
        if _fast3(x, Co):
            y = conv3x3_s1(x, _pack_fwd(w, "s1", x.dtype), b, res=h)[0]
        else:
            y = _empty((B, Fi, Ti, Co), x.dtype, x.device)
            conv_gemm(x, _pack_fwd(w, "s1", x.dtype), b, y, gf, res=h)
        ctx.save_for_backward(x, w)
        ctx.bias_param = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx, dw = _conv_backward("s1", x, w, dy, ctx.needs_input_grad[0])
        Co = dy.shape[3]
        with _lane(dy.device, dy, on=_lane_ok(ctx.bias_param)):
            db = colsum(dy, dy.numel() // Co, Co)
        if ctx.slot is not None and dx is not None:
            ctx.slot.put(dx)                 # x's other consumer (conv1[0]) adds it inside its data-gradient conv
            dx = None
        return dx, dw, db, dy, None


class Conv3x3(torch.autograd.Function):
    """Bias-free 3x3 stride-1 pad-1 convolution of the ResNet-18 lip encoder (VideoEncoders/resnet.py:26-29, conv3x3)
    on channels-last frames: x [N, C, H, W] in channels-last memory (viewed as [N, H, W, C]), w [Co, Ci, 3, 3] fp32.
    Images of >= 400 pixels (layer1: 22x22) run on the 16x16-pixel tile kernels, smaller ones (11x11, 6x6, 3x3) as
    implicit GEMMs over all frames (forward, data gradient with rotated taps, gathered weight gradient).  Measured
    against MIOpen's igemm kernels at 1600 frames, forward + backward per conv: 464 vs 562 us at 64 @ 22x22, 342 vs
    426 at 128 @ 11x11, 435 vs 426 at 256 @ 6x6, 407 vs 487 at 512 @ 3x3; the three stride-2 convs stay MIOpen's
    (ours: 324-365 vs 285-306 us)."""

    @staticmethod
    def forward(ctx, x, w, stride=1, slot=None):
        ctx.slot = slot
        xc = x.permute(0, 2, 3, 1)
        if not xc.is_contiguous():
            xc = xc.contiguous()
        B, Fi, Ti, Ci = xc.shape
        Co = w.shape[0]
        kind = "s1" if stride == 1 else "down"              # stride 2: implicit GEMM forward, four-parity data gradient
        tile = kind == "s1" and Fi * Ti >= 400 and _fast3(xc, Co)
        wp = _pack_fwd(w, kind, xc.dtype)
        if tile:
            y = conv3x3_s1(xc, wp, None)[0]
        else:
            Fo, To, gf, _, _ = _conv_geoms(kind, B, Fi, Ti, Ci, Co)
            y = _empty((B, Fo, To, Co), xc.dtype, xc.device)
            conv_gemm(xc, wp, None, y, gf)
        ctx.save_for_backward(xc, w)
        ctx.tile, ctx.kind = tile, kind
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        xc, w = ctx.saved_tensors
        dyc = dy.permute(0, 2, 3, 1)
        if not dyc.is_contiguous():
            dyc = dyc.contiguous()
        res = res00 = None
        if ctx.slot is not None and ctx.needs_input_grad[0]:
            g, k = ctx.slot.take()           # the partial gradient of x's other consumer, if it ran first (GradSlot)
            if k == "even":                  # Conv1x1s2 (downsample branch): compact, lands on the even pixels
                res00 = g
            elif g is not None:              # BatchNormAct's residual input: shape of x (NCHW view, channels-last)
                res = g.permute(0, 2, 3, 1)
                if not res.is_contiguous():
                    res = res.contiguous()
        if res00 is not None and ctx.kind != "down":
            raise DttError("Conv3x3: an even-pixel partial gradient needs the stride-2 form")
        dx, dw = _conv_backward(ctx.kind, xc, w, dyc, ctx.needs_input_grad[0], tile=ctx.tile, phases=True, res=res,
                                res00=res00)
        if dx is not None:
            dx = dx.permute(0, 3, 1, 2)
            if ctx.slot is not None and res is None and res00 is None and ctx.kind == "down":
                # ran BEFORE the downsample branch's backward (engine order): park the full gradient, Conv1x1s2
                # adds its own part to the even pixels
                ctx.slot.put(dx, "full")
                dx = None
        return dx, dw, None, None


class Conv1x1s2(torch.autograd.Function):
    """Bias-free 1x1 stride-2 convolution (the downsample branch of a ResNet block, VideoEncoders/resnet.py:31-33
    downsample_basic_block) on channels-last frames: x [N, Ci, H, W] (channels-last memory), w [Co, Ci, 1, 1].
    Forward and weight gradient gather every second pixel through dtt_gemm's general conv geometry (kernel width 1,
    no padding); the data gradient is a plain GEMM scattered to the even pixels of a zeroed tensor."""

    @staticmethod
    def forward(ctx, x, w, slot=None):
        ctx.slot = slot
        xc = x.permute(0, 2, 3, 1)
        if not xc.is_contiguous():
            xc = xc.contiguous()
        B, Fi, Ti, Ci = xc.shape
        Co = w.shape[0]
        Fo, To = (Fi - 1) // 2 + 1, (Ti - 1) // 2 + 1
        geom = dict(cB=B, cFi=Fi, cTi=Ti, cCi=Ci, cFo=Fo, cTo=To, cstride=2, ctrans=0, cKW=1, cPadF=0, cPadT=0)
        wp = w.view(Co, Ci).to(xc.dtype)
        y = _empty((B, Fo, To, Co), xc.dtype, xc.device)
        gemm_raw(A=xc, B=wp, C=y, M=B * Fo * To, N=Co, K=Ci, a_ms=0, a_ks=1, b_ks=1, b_ns=Ci, c_ms=Co, c_ns=1, dtype=_dt(xc),
                 a_mode=1, **geom)
        ctx.save_for_backward(xc, w)
        ctx.geom = geom
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        xc, w = ctx.saved_tensors
        g = ctx.geom
        B, Fi, Ti, Ci = xc.shape
        Co = w.shape[0]
        dyc = dy.permute(0, 2, 3, 1)
        if not dyc.is_contiguous():
            dyc = dyc.contiguous()
        P = B * g["cFo"] * g["cTo"]
        dwp = _zeros_f32((Co, Ci), xc.device)
        with _lane(xc.device, dyc, xc, on=_lane_ok(w)):
            gemm_raw(A=dyc, B=xc, C=dwp, M=Co, N=Ci, K=P, splitk=_splitk(Co, Ci, P), a_ms=1, a_ks=Co, b_ks=Ci, b_ns=1, c_ms=Ci,
                     c_ns=1, dtype=_dt(xc), atomic=1, out_f32=1, b_mode=1, **g)
        dx = None
        if ctx.needs_input_grad[0]:
            t = _empty((P, Ci), xc.dtype, xc.device)
            wq = w.view(Co, Ci).to(xc.dtype)                         # [Co][Ci]: B(k = co, n = ci) row-contiguous
            gemm_raw(A=dyc, B=wq, C=t, M=P, N=Ci, K=Co, a_ms=Co, a_ks=1, b_ks=Ci, b_ns=1, c_ms=Ci, c_ns=1, dtype=_dt(xc))
            t = t.view(B, g["cFo"], g["cTo"], Ci)
            full, kind = ctx.slot.take() if ctx.slot is not None else (None, None)
            if full is not None:
                # the 3x3 stride-2 conv over the same x already ran: add this branch to its even pixels, report the sum
                dxc = full.permute(0, 2, 3, 1)
                dxc[:, ::2, ::2] += t
                dx = full
            elif ctx.slot is not None:
                # normal order: the 3x3 stride-2 conv's four-parity data gradient takes t as the residual of its
                # even-even GEMM (no 180 MB zero fill + strided scatter + add per block)
                ctx.slot.put(t, "even")
                dx = None
            else:
                dx = torch.zeros_like(xc)
                dx[:, ::2, ::2] = t
                dx = dx.permute(0, 3, 1, 2)
        return dx, dwp.view(Co, Ci, 1, 1), None


class FrontConv(torch.autograd.Function):
    """The lip front end's Conv3d(1, Co, (kt, 7, 7), stride (1, 2, 2), padding (kt // 2, 3, 3), bias=False)
    (resnet_video_encoder.py:33-37) as implicit GEMMs over the stacked-frame tensor of dtt_stack_frames:
    x5p [frames, H, 3 + W + 5, 8] (kt <= 8 neighbouring frames as channels, zero columns around every row), w
    [Co, 1, kt, 7, 7] fp32 -> y [frames, H/2, W/2, Co].  A kernel row (8 pixels of 8 channels, the eighth pixel and
    the channels >= kt with zero weights) is 128 contiguous bytes of x5p: forward K = 7 rows x 64, weight gradient
    N = 7 x 64 with the pixels as K (dtt_gemm's general conv geometry).  No input gradient (the clip is data).
    MIOpen: 707 us forward + 777 us weight gradient at 1600 frames of 88 x 88."""

    @staticmethod
    def forward(ctx, x5p, w, H, W):
        NF, Hx, Wp, Cc = x5p.shape
        Co, _, kt, KH, KWr = w.shape
        Ho, Wo = (H + 6 - KH) // 2 + 1, (W + 6 - KWr) // 2 + 1
        if not (Hx == H and Cc == 8 and KWr <= 8 and kt <= 8 and Wp >= 2 * (Wo - 1) + 8 and Wp >= W + 3):
            raise DttError("FrontConv: input is not a dtt_stack_frames tensor padded by (3, >= 5) columns")
        dt = x5p.dtype
        wp = torch.zeros((Co, KH, 8, 8), dtype=dt, device=w.device)
        wp[:, :, :KWr, :kt] = w[:, 0].permute(0, 2, 3, 1)
        geom = dict(cB=NF, cFi=H, cTi=Wp, cCi=8, cFo=Ho, cTo=Wo, cstride=2, ctrans=0, cKW=8, cPadF=3, cPadT=0)
        y = _empty((NF, Ho, Wo, Co), dt, x5p.device)
        gemm_raw(A=x5p, B=wp, C=y, M=NF * Ho * Wo, N=Co, K=KH * 64, a_ms=0, a_ks=1, b_ks=1, b_ns=KH * 64, c_ms=Co, c_ns=1,
                 dtype=_dt(x5p), a_mode=1, **geom)
        ctx.save_for_backward(x5p)
        ctx.geom, ctx.wshape, ctx.lane_params = geom, tuple(w.shape), (w,)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x5p,) = ctx.saved_tensors
        Co, _, kt, KH, KWr = ctx.wshape
        g = ctx.geom
        dy = dy.contiguous()
        P, N = g["cB"] * g["cFo"] * g["cTo"], KH * 64
        dwp = _zeros_f32((Co, N), dy.device)
        tiles = -(-Co // 64) * -(-N // 128)
        with _lane(dy.device, dy, x5p, dwp, on=_lane_ok(*ctx.lane_params)):
            gemm_raw(A=dy, B=x5p, C=dwp, M=Co, N=N, K=P, splitk=max(1, min(256, 512 // tiles)), a_ms=1, a_ks=Co, b_ks=N,
                     b_ns=1, c_ms=N, c_ns=1, dtype=_dt(dy), atomic=1, out_f32=1, b_mode=1, **g)
            dw = dwp.view(Co, KH, 8, 8)[:, :, :KWr, :kt].permute(0, 3, 1, 2).unsqueeze(1).contiguous()
        return None, dw, None, None


# --------------------------------------------------------------------------- #
# TDF: two Linears on the frequency axis + residual (block_tfc_tdf.py:23-26,43-44)
# --------------------------------------------------------------------------- #
class TDF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        B, Fq, T, Cc = x.shape
        J = w1.shape[0]
        N = T * Cc
        dt = x.dtype
        Fp = (Fq + 7) // 8 * 8
        def pad_w1():
            t = torch.zeros(J, Fp, dtype=dt, device=x.device)
            t[:, :Fq] = w1
            return t
        w1p = packed(_wkey("tdf1", dt, w1), pad_w1, ((J, Fp), dt, lambda out: [(out[:, :Fq], w1)]))
        w2p = packed(_wkey("tdf2", dt, w2), lambda: w2.to(dt).contiguous(), (tuple(w2.shape), dt, lambda out: [(out, w2)]))   # [F][J]
        hid = _empty((B, J, N), dt, x.device)
        gemm_raw(A=w1p, B=x, C=hid, bias=b1, M=J, N=N, K=Fq, batch=B, a_ms=Fp, a_ks=1, a_bs=0, b_ks=N, b_ns=1,
                 b_bs=Fq * N, c_ms=N, c_ns=1, c_bs=J * N, dtype=_dt(x), bias_mode=2)
        out = torch.empty_like(x)
        gemm_raw(A=w2p, B=hid, C=out, bias=b2, R=x, M=Fq, N=N, K=J, batch=B, a_ms=J, a_ks=1, a_bs=0, b_ks=N,
                 b_ns=1, b_bs=J * N, c_ms=N, c_ns=1, c_bs=Fq * N, r_ms=N, r_ns=1, r_bs=Fq * N, dtype=_dt(x),
                 bias_mode=2)
        ctx.save_for_backward(x, hid, w1p, w2p)
        ctx.lane_params = (w1, b1, w2, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, hid, w1p, w2p = ctx.saved_tensors
        dout = dout.contiguous()
        B, Fq, T, Cc = x.shape
        J, Fp = w1p.shape
        N = T * Cc
        d = _dt(x)
        dev = x.device
        dhid = torch.empty_like(hid)
        gemm_raw(A=w2p, B=dout, C=dhid, M=J, N=N, K=Fq, batch=B, a_ms=1, a_ks=J, a_bs=0, b_ks=N, b_ns=1,
                 b_bs=Fq * N, c_ms=N, c_ns=1, c_bs=J * N, dtype=d)
        dx = torch.empty_like(x)
        dw2, dw1, db2, db1 = _zeros_f32((Fq, J,), dev), _zeros_f32((J, Fq,), dev), _zeros_f32((Fq,), dev), _zeros_f32((J,), dev)
        ones = torch.ones(N, dtype=x.dtype, device=dev)
        with _lane(dev, dout, hid, dhid, x, ones, on=_lane_ok(*ctx.lane_params)):
            gemm_raw(A=dout, B=hid, C=dw2, M=Fq, N=J, K=N, batch=B, splitk=_splitk(Fq, J, N, B), a_ms=N, a_ks=1, a_bs=Fq * N,
                     b_ks=1, b_ns=N, b_bs=J * N, c_ms=J, c_ns=1, c_bs=0, dtype=d, atomic=1, out_f32=1)
            gemm_raw(A=dhid, B=x, C=dw1, M=J, N=Fq, K=N, batch=B, splitk=_splitk(J, Fq, N, B), a_ms=N, a_ks=1, a_bs=J * N,
                     b_ks=1, b_ns=N, b_bs=Fq * N, c_ms=Fq, c_ns=1, c_bs=0, dtype=d, atomic=1, out_f32=1)
            gemm_raw(A=dout, B=ones, C=db2, M=Fq, N=1, K=N, batch=B, a_ms=N, a_ks=1, a_bs=Fq * N, b_ks=1, b_ns=0,
                     b_bs=0, c_ms=1, c_ns=1, c_bs=0, dtype=d, atomic=1, out_f32=1)
            gemm_raw(A=dhid, B=ones, C=db1, M=J, N=1, K=N, batch=B, a_ms=N, a_ks=1, a_bs=J * N, b_ks=1, b_ns=0,
                     b_bs=0, c_ms=1, c_ns=1, c_bs=0, dtype=d, atomic=1, out_f32=1)
        gemm_raw(A=w1p, B=dhid, C=dx, R=dout, M=Fq, N=N, K=J, batch=B, a_ms=1, a_ks=Fp, a_bs=0, b_ks=N, b_ns=1,
                 b_bs=J * N, c_ms=N, c_ns=1, c_bs=Fq * N, r_ms=N, r_ns=1, r_bs=Fq * N, dtype=d)
        return dx, dw1, db1, dw2, db2


# --------------------------------------------------------------------------- #
# TC_FC_RNN: GroupNorm -> BiLSTM -> Linear -> + x   (latent.py:77-104)
# --------------------------------------------------------------------------- #
def _pack_whh_fwd(w, dt):
    """[4H, H] -> MFMA B-fragment order [4H/32][H/16][64 lanes][8]."""
    G4, H = w.shape
    return _cast_perm(w.view(G4 // 32, 32, H // 16, 2, 8), (0, 2, 3, 1, 4), dt)


def _pack_whh_bwd(w, dt):
    """[4H, H] -> fragments of B[k][n] = W_hh[k][n] for dh = dG * W_hh: [H/32][4H/16][64][8]."""
    G4, H = w.shape
    return _cast_perm(w.view(G4 // 16, 2, 8, H // 32, 32), (3, 0, 1, 4, 2), dt)


def _pack_pair(tag, wa, wb, dt, fwd):
    """Both directions' weights in MFMA fragment order (_pack_whh_fwd / _pack_whh_bwd), stacked."""
    G4, H = wa.shape
    if fwd:
        vw = lambda w: w.view(G4 // 32, 32, H // 16, 2, 8).permute(0, 2, 3, 1, 4)
        fn = _pack_whh_fwd
    else:
        vw = lambda w: w.view(G4 // 16, 2, 8, H // 32, 32).permute(3, 0, 1, 4, 2)
        fn = _pack_whh_bwd
    shape = (2,) + tuple(vw(wa).shape)
    return packed(_wkey(tag, dt, wa, wb), lambda: torch.stack([fn(wa, dt), fn(wb, dt)]).contiguous(),
                  (shape, dt, lambda out: [(out[0], vw(wa)), (out[1], vw(wb))]))


_DXN_T = os.environ.get("DTT_DXN_T", "1") != "0"      # 0: dxn = dG W_ih with W_ih as stored (A/B switch)


def _unblock(t, W, N, L, heads):
    """blocked [2][nblk][L][W/H..] activation layout of the fused LSTM kernels -> row-major [R][2][W]
    (debug / transition helper: index q = lh*16 + r of the last axis is local sequence
    (r&3) + 8*(r>>2) + 4*lh)."""
    nblk = t.shape[1]
    q = torch.arange(32, device=t.device)
    lh, r = q // 16, q % 16
    nl = (r & 3) + 8 * (r >> 2) + 4 * lh
    inv = torch.empty_like(nl); inv[nl] = q                 # position of local sequence nl in the lane order
    v = t.view(2, nblk, L, W, 32)[..., inv]                 # [2][nblk][L][W][nl]
    v = v.permute(1, 4, 2, 0, 3).reshape(nblk * 32, L, 2, W)[:N]         # [n][L][2][W]
    v = v.view(N // heads, heads, L, 2, W).permute(0, 2, 1, 3, 4).reshape(N * L, 2, W)
    return v.contiguous()


def _cluster_geometry(N, H, d):
    """(U, G, S) of the weight-stationary cluster kernels for this shape on this GPU, or None."""
    U, G, S = C.c_int(0), C.c_int(0), C.c_int(0)
    if lib().dtt_lstm_cluster_geometry(N, H, d, C.byref(U), C.byref(G), C.byref(S)) != 0:
        return None
    return U.value, G.value, S.value


_HANDOFF = True


def handoff_kernels(enabled=True):
    """Allow / forbid the recurrent kernels whose CUs hand data to each other inside one launch (lstm_rs.hip,
    lstm_cluster.hip); with False `auto` selects the streaming kernels (lstm.hip), which need no co-residency.
    They are recorded into HIP graphs by default since their counters are reset by a kernel (DESIGN finding 34:
    as memset NODES those resets made replays after a host sync return wrong gradients);
    GraphedTrainStep(handoff_kernels=False) / DTT_GRAPH_HANDOFF=0 keeps them out of a recorded step.
    Process-wide, like the pack plan; the setting at FORWARD time decides for the backward of that forward too."""
    global _HANDOFF
    _HANDOFF = bool(enabled)


def handoff_spin_limit(polls=0):
    """Polls (~1 us each) a hand-off wait makes before it gives up and raises the status words; 0 = the default of
    2^23 (~8 s), long enough for a data-parallel job's rank skew (a collective that waits for a peer can hold CUs
    the launch needs: first steps, graph captures on other ranks).  A time-out raises DttError at the next poll /
    check; it never goes unreported."""
    check(lib().dtt_lstm_handoff_spin_limit(C.c_uint(int(polls))), "dtt_lstm_handoff_spin_limit")


# --------------------------------------------------------------------------- #
# Hand-off health: the product reads the status words
# --------------------------------------------------------------------------- #
class _HandoffWatch:
    """Per GPU.  A hand-off launch (lstm_rs.hip / lstm_cluster.hip) whose workgroups are not all co-resident -- a
    CU-masked partition, a foreign kernel (RCCL beside backward) holding a CU, two processes on one card -- times
    out after ~0.5 s, raises a sticky per-GPU status word and finishes on stale tiles: that step's activations and
    gradients are garbage.  `poll` (non-blocking: loss wrappers, optimizer steps, graph replays, the reducer's
    finish) folds the sticky words into a device flag and copies it to pinned host memory behind the work
    already queued; the NEXT poll that finds the copy complete raises.  `check` (blocking: metrics, checkpoint
    save, bench end, anything that synchronises anyway) does both at once."""

    def __init__(self, dev):
        self.dev = dev
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event = None

    def fold(self, clear=0):
        st = stream_ptr()
        check(lib().dtt_lstm_rs_status_fold(ptr(self.flag), clear, st), "dtt_lstm_rs_status_fold")
        check(lib().dtt_lstm_cluster_status_fold(ptr(self.flag), clear, st), "dtt_lstm_cluster_status_fold")

    def fail(self, bits):
        with torch.cuda.device(self.dev):
            self.fold(clear=1)                       # re-arm: the caller may switch kernels and go on
            self.flag.zero_()
        self.host.zero_()
        self.event = None
        fam = " and ".join(n for b, n in ((1, "register-stationary (lstm_rs)"), (2, "LDS-cluster (lstm_cluster)")) if bits & b)
        raise DttError(
            f"a {fam} BiLSTM launch on {self.dev} timed out waiting for a peer workgroup: its hidden states / gate "
            "gradients and everything computed from them since the last successful check are INVALID (discard the "
            "step).  These kernels need all of their workgroups co-resident, one per CU: a CU-masked partition, "
            "another process on the card or a foreign kernel holding CUs breaks that.  Use ops.handoff_kernels(False) "
            "or DTT_LSTM=stream for the streaming kernels, which need no co-residency.")


_WATCH = {}


def _watch(dev=None):
    if dev is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    w = _WATCH.get(key)
    if w is None:
        w = _WATCH[key] = _HandoffWatch(torch.device("cuda", key))
    return w


_HANDOFF_USED = set()        # device indices with hand-off launches issued since their last fold


def _dev_index(dev):
    return dev.index if dev is not None and dev.index is not None else torch.cuda.current_device()


def _note_handoff(dev):
    """Called at every hand-off launch site (also while a graph is being recorded: touches no tensor)."""
    _HANDOFF_USED.add(_dev_index(dev))


def handoff_poll(dev=None, used=None):
    """Non-blocking health check of the hand-off LSTM kernels (see _HandoffWatch).  Raises DttError when a copy
    queued by an EARLIER poll has arrived with a time-out in it; queues a new fold + copy when hand-off launches
    were issued since (`used=True`: a graph replay holding such launches).  No-op while a stream is capturing."""
    key = _dev_index(dev)
    if used:
        _HANDOFF_USED.add(key)
    if torch.cuda.is_current_stream_capturing() or (key not in _WATCH and key not in _HANDOFF_USED):
        return
    w = _watch(torch.device("cuda", key))
    if w.event is not None and w.event.query():
        w.event = None
        bits = int(w.host[0])
        if bits:
            w.fail(bits)
    if w.event is None and key in _HANDOFF_USED:
        _HANDOFF_USED.discard(key)
        with torch.cuda.device(w.dev):
            w.fold()
            w.host.copy_(w.flag, non_blocking=True)
            w.event = torch.cuda.Event()
            w.event.record()


def handoff_check(dev=None):
    """Blocking form: everything queued on the current stream so far is covered.  Call it where the host
    synchronises anyway (metrics' .cpu(), checkpoint save, end of a benchmark)."""
    if torch.cuda.is_current_stream_capturing():
        return
    w = _watch(dev)
    _HANDOFF_USED.discard(_dev_index(w.dev))
    with torch.cuda.device(w.dev):
        w.fold()
        bits = int(w.flag.item())
    w.event = None
    if bits:
        w.fail(bits)


def _lstm_mode():
    """DTT_LSTM = auto | rs | cluster | stream (A/B switch between the recurrent kernel families)."""
    mode = os.environ.get("DTT_LSTM", "auto")
    return "stream" if (mode == "auto" and not _HANDOFF) else mode


def _rs_ok(N, L, H, d, heads, mode):
    """Register-stationary kernels (lstm_rs.hip): bf16, H = 512 (auto) or H in {256, 512} (DTT_LSTM=rs), and N * L
    within the kernels' 32-bit offsets (beyond: the cluster / streaming kernels)."""
    if mode not in ("auto", "rs") or heads & (heads - 1) or heads > 32:
        return False
    if mode == "auto" and H != 512:
        return False
    return lib().dtt_lstm_rs_covers(N, L, H, d) == 0


class DualPathRNN(torch.autograd.Function):
    """One TC_FC_RNN over the time axis (along_f=False) or the frequency axis
    (along_f=True) of x [B,F,T,C], heads folded into the sequence batch.
    The frequency variant returns its result in [B,T,F,C] memory order, which
    the caller re-views as [B,F,T,C]: that is the reference's merge_heads
    (latent.py:68-74), which views without permuting back."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, fc_w, fc_b, heads,
                along_f):
        B, Fq, T, Cc = x.shape
        Cp = Cc // heads
        H = w_hh.shape[1]
        dt, dev, d = x.dtype, x.device, _dt(x)
        if along_f:
            nslab, NI, L, OS, IS, RS = B * T, T, Fq, Fq * T * Cc, Cc, T * Cc
        else:
            nslab, NI, L, OS, IS, RS = B * Fq, 1, T, T * Cc, 0, Cc
        gs = 16 if Cp % 16 == 0 else Cp
        ng = Cc // gs
        R = nslab * L * heads
        xn = _empty((nslab, L, Cc), dt, dev)
        xperm = _empty((nslab, L, Cc), dt, dev) if along_f else None
        gstats = _empty((nslab, ng, 2), F32, dev)
        geo = (nslab, NI, L, Cc, Cp, gs, C.c_int64(OS), C.c_int64(IS), C.c_int64(RS))
        check(lib().dtt_gn_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(xn), ptr(xperm), ptr(gstats), *geo,
                               C.c_float(1e-5), d, stream_ptr()), "dtt_gn_fwd")
        wih = packed(_wkey("wih", dt, w_ih, w_ih_r), lambda: torch.cat([w_ih, w_ih_r], 0).to(dt).contiguous(),
                     ((8 * H, Cp), dt, lambda out: [(out[:4 * H], w_ih), (out[4 * H:], w_ih_r)]))    # [8H][Cp]
        bsum = packed(_wkey("bsum", F32, b_ih, b_hh, b_ih_r, b_hh_r),
                      lambda: torch.cat([b_ih + b_hh, b_ih_r + b_hh_r]).contiguous())
        y = _empty((R, 2 * H), dt, dev)
        need = any(ctx.needs_input_grad)
        N = nslab * heads
        mode = _lstm_mode()
        rs = _rs_ok(N, L, H, d, heads, mode)
        blocked = dt == torch.bfloat16 and H == 256 and Cp == 128 and mode not in ("cluster", "rs")
        geom = None if (rs or blocked or mode == "stream") else _cluster_geometry(N, H, d)
        if rs:
            # recurrent weights resident in the register file of a cluster of CUs (lstm_rs.hip): input projection
            # by GEMM, h exchanged through y, activations saved as (4 units, 1 sequence) granules
            xg = _empty((R, 8 * H), dt, dev)
            gemm_raw(A=xn, B=wih, C=xg, bias=bsum, M=R, N=8 * H, K=Cp, a_ms=Cp, a_ks=1, b_ks=1, b_ns=Cp,
                     c_ms=8 * H, c_ns=1, dtype=d, bias_mode=1)
            NT = (N + 31) // 32
            whh = packed(_wkey("whh", dt, w_hh, w_hh_r), lambda: torch.stack([w_hh, w_hh_r]).to(dt).contiguous(),
                         ((2, 4 * H, H), dt, lambda out: [(out[0], w_hh), (out[1], w_hh_r)]))   # [2][4H][H]
            gates = _empty((2, NT, L, 4 * H * 32), dt, dev) if need else None
            cst = _empty((2, NT, L, H * 32), F32, dev)
            sync = _empty((lib().dtt_lstm_rs_sync_words(N, H),), torch.int32, dev)
            check(lib().dtt_lstm_rs_fwd(ptr(xg), ptr(whh), ptr(y), ptr(gates), ptr(cst), ptr(sync), N, L, heads, H,
                                        d, stream_ptr()), "dtt_lstm_rs_fwd")
            _note_handoff(dev)
        elif geom is not None:
            # weight-stationary recurrence over a cluster of CUs (lstm_cluster.hip): input projection by GEMM,
            # W_hh resident in LDS, h exchanged through y; activations saved in the blocked layout
            xg = _empty((R, 8 * H), dt, dev)
            gemm_raw(A=xn, B=wih, C=xg, bias=bsum, M=R, N=8 * H, K=Cp, a_ms=Cp, a_ks=1, b_ks=1, b_ns=Cp,
                     c_ms=8 * H, c_ns=1, dtype=d, bias_mode=1)
            NT = (N + 31) // 32
            whh = packed(_wkey("whh", dt, w_hh, w_hh_r), lambda: torch.stack([w_hh, w_hh_r]).to(dt).contiguous(),
                         ((2, 4 * H, H), dt, lambda out: [(out[0], w_hh), (out[1], w_hh_r)]))   # [2][4H][H]
            gates = _empty((2, NT, L, 4 * H * 32), dt, dev) if need else None
            cst = _empty((2, NT, L, H * 32), F32, dev)
            sync = _empty((lib().dtt_lstm_cluster_sync_words(geom[2]),), torch.int32, dev)
            check(lib().dtt_lstm_cluster_fwd(ptr(xg), ptr(whh), ptr(y), ptr(gates), ptr(cst), ptr(sync), N, L, heads, H,
                                             d, stream_ptr()), "dtt_lstm_cluster_fwd")
            _note_handoff(dev)
        elif blocked:
            # fused input projection, activations saved in the kernels' private blocked layout
            nblk = (N + 31) // 32
            wp = _pack_pair("whhp", w_hh, w_hh_r, dt, True)
            wihp = _pack_pair("wihp", w_ih, w_ih_r, dt, True)
            gates = _empty((2, nblk, L, 4 * H * 32), dt, dev) if need else None
            cst = _empty((2, nblk, L, H * 32), F32, dev) if need else None
            rc = lib().dtt_lstm_fwd_fused(ptr(xn), ptr(wihp), ptr(wp), ptr(bsum), ptr(y), ptr(gates), ptr(cst), N, L,
                                          heads, H, Cp, d, stream_ptr())
            if rc == 1:
                raise DttError("dtt_lstm_fwd_fused refused a shape the host side selected")
            check(rc, "dtt_lstm_fwd_fused")
        else:
            # input projection, both directions: xg [R][2][4H]
            xg = _empty((R, 8 * H), dt, dev)
            gemm_raw(A=xn, B=wih, C=xg, bias=bsum, M=R, N=8 * H, K=Cp, a_ms=Cp, a_ks=1, b_ks=1, b_ns=Cp,
                     c_ms=8 * H, c_ns=1, dtype=d, bias_mode=1)
            gates = _empty((R, 2, 4 * H), dt, dev) if need else None
            cst = _empty((R, 2, H), F32, dev)            # always: the cell state round-trips through it
            wp = _pack_pair("whhp", w_hh, w_hh_r, dt, True)
            check(lib().dtt_lstm_fwd(ptr(xg), ptr(wp), ptr(y), ptr(gates), ptr(cst), N, L, heads, H, d, stream_ptr()),
                  "dtt_lstm_fwd")
        fcw = packed(_wkey("fcw", dt, fc_w), lambda: fc_w.to(dt).contiguous(),
                     (tuple(fc_w.shape), dt, lambda out: [(out, fc_w)]))                     # [Cp][4Cp]
        res = xperm if along_f else x
        out = _empty((nslab, L, Cc), dt, dev)
        gemm_raw(A=y, B=fcw, C=out, bias=fc_b, R=res, M=R, N=Cp, K=2 * H, a_ms=2 * H, a_ks=1, b_ks=1, b_ns=2 * H,
                 c_ms=Cp, c_ns=1, r_ms=Cp, r_ns=1, dtype=d, bias_mode=1)
        ctx.save_for_backward(x, gamma, xn, gstats, wih, w_hh, w_hh_r, y, gates, cst, fcw)
        ctx.w_ih_pair = (w_ih, w_ih_r)                 # (parameters: sources of the transposed pack in backward)
        ctx.lane_params = (w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r, fc_w, fc_b)
        ctx.cfg = (heads, along_f, geo, gs, blocked, geom, rs)
        # along_f: memory order is [B,T,F,C]; the view below is the reference's merge_heads quirk
        return out.view(B, Fq, T, Cc)

    @staticmethod
    def backward(ctx, dout):
        x, gamma, xn, gstats, wih, w_hh, w_hh_r, y, gates, cst, fcw = ctx.saved_tensors
        heads, along_f, geo, gs, blocked, geom, rs = ctx.cfg
        nslab, NI, L, Cc, Cp = geo[0], geo[1], geo[2], geo[3], geo[4]
        dt, dev, d = x.dtype, x.device, _dt(x)
        H = w_hh.shape[1]
        R = nslab * L * heads
        if blocked and os.environ.get("DTT_LSTM_BWD_UNBLOCK") == "1":      # debug: old kernel on converted layouts
            gates, cst = _unblock(gates, 4 * H, nslab * heads, L, heads), _unblock(cst, H, nslab * heads, L, heads)
            blocked = False
        dout = dout.contiguous()                       # memory order == rows (outer, step, head)
        # Linear backward
        dy = _empty((R, 2 * H), dt, dev)
        gemm_raw(A=dout, B=fcw, C=dy, M=R, N=2 * H, K=Cp, a_ms=Cp, a_ks=1, b_ks=2 * H, b_ns=1, c_ms=2 * H, c_ns=1,
                 dtype=d)
        dfcw = _zeros_f32((Cp, 2 * H,), dev)
        lane_on = _lane_ok(*ctx.lane_params)
        with _lane(dev, dout, y, on=lane_on):
            gemm_raw(A=dout, B=y, C=dfcw, M=Cp, N=2 * H, K=R, splitk=_splitk(Cp, 2 * H, R), a_ms=1, a_ks=Cp, b_ks=2 * H,
                     b_ns=1, c_ms=2 * H, c_ns=1, dtype=d, atomic=1, out_f32=1)
            dfcb = colsum(dout, R, Cp)
        # recurrent backward
        dG = _empty((R, 2, 4 * H), dt, dev)
        dbias = None
        if rs:
            N = nslab * heads
            NT = (N + 31) // 32
            whhT = packed(_wkey("whhT", dt, w_hh, w_hh_r), lambda: torch.stack([w_hh.t(), w_hh_r.t()]).to(dt).contiguous(),
                          ((2, H, 4 * H), dt, lambda out: [(out[0], w_hh.t()), (out[1], w_hh_r.t())]))   # [2][H][4H]
            dcbuf = _empty((2, NT, H * 32), F32, dev)
            dbias = _zeros_f32((2 * 4 * H,), dev)
            sync = _empty((lib().dtt_lstm_rs_sync_words(N, H),), torch.int32, dev)
            check(lib().dtt_lstm_rs_bwd(ptr(dy), ptr(gates), ptr(cst), ptr(whhT), ptr(dG), ptr(dcbuf), ptr(dbias),
                                        ptr(sync), N, L, heads, H, d, stream_ptr()), "dtt_lstm_rs_bwd")
            _note_handoff(dev)
        elif geom is not None:
            N = nslab * heads
            NT = (N + 31) // 32
            whhT = packed(_wkey("whhT", dt, w_hh, w_hh_r), lambda: torch.stack([w_hh.t(), w_hh_r.t()]).to(dt).contiguous(),
                          ((2, H, 4 * H), dt, lambda out: [(out[0], w_hh.t()), (out[1], w_hh_r.t())]))   # [2][H][4H]
            dcbuf = _empty((2, NT, H * 32), F32, dev)
            dbias = _zeros_f32((2 * 4 * H,), dev)
            sync = _empty((lib().dtt_lstm_cluster_sync_words(geom[2]),), torch.int32, dev)
            check(lib().dtt_lstm_cluster_bwd(ptr(dy), ptr(gates), ptr(cst), ptr(whhT), ptr(dG), ptr(dcbuf), ptr(dbias),
                                             ptr(sync), N, L, heads, H, d, stream_ptr()), "dtt_lstm_cluster_bwd")
            _note_handoff(dev)
        elif blocked:
            wpt = _pack_pair("whhpt", w_hh, w_hh_r, dt, False)
            dbias = _zeros_f32((2 * 4 * H,), dev)           # bias gradient accumulated inside the kernel
            check(lib().dtt_lstm_bwd_blk(ptr(dy), ptr(gates), ptr(cst), ptr(wpt), ptr(dG), ptr(dbias), nslab * heads,
                                         L, heads, H, d, stream_ptr()), "dtt_lstm_bwd_blk")
        else:
            wpt = _pack_pair("whhpt", w_hh, w_hh_r, dt, False)
            check(lib().dtt_lstm_bwd(ptr(dy), ptr(gates), ptr(cst), ptr(wpt), ptr(dG), nslab * heads, L, heads, H, d,
                                     stream_ptr()), "dtt_lstm_bwd")
        G4 = 4 * H
        dwhh = [_zeros_f32((G4, H,), dev), _zeros_f32((G4, H,), dev)]
        dwih = _zeros_f32((2 * G4, Cp,), dev)
        with _lane(dev, dG, y, xn, on=lane_on):
            for di, (shift, edge) in enumerate(((-heads, 0), (heads, L - 1))):
                gemm_raw(A=dG[:, di], B=y[:, di * H:], C=dwhh[di], M=G4, N=H, K=R, splitk=_splitk(G4, H, R), a_ms=1, a_ks=2 * G4,
                         b_ks=2 * H, b_ns=1, c_ms=H, c_ns=1, dtype=d, atomic=1, out_f32=1, b_mode=2, sh_shift=shift,
                         sh_heads=heads, sh_L=L, sh_edge=edge)
            gemm_raw(A=dG, B=xn, C=dwih, M=2 * G4, N=Cp, K=R, splitk=_splitk(2 * G4, Cp, R), a_ms=1, a_ks=2 * G4, b_ks=Cp,
                     b_ns=1, c_ms=Cp, c_ns=1, dtype=d, atomic=1, out_f32=1)
        if dbias is None:           # (on the launch stream: each half is returned for two parameters, which autograd clones)
            dbias = colsum(dG, R, 2 * G4)
        dxn = _empty((nslab, L, Cc), dt, dev)
        if _DXN_T and d == 1 and Cp % 8 == 0 and Cp >= 256:
            # W_ih transposed, [Cp][8H]: both operands k-contiguous, which is what the 256 x 256 LDS-DMA tiles take
            # (33 280 x 256 x 4096: 125 us against 154 for the 128 x 128 tiles of the n-contiguous form; in the step,
            # beside the weight-gradient products on their side stream, -0.08 ms: DTT_DXN_T=0 is the A/B switch)
            w_ih, w_ih_r = ctx.w_ih_pair
            wihT = packed(_wkey("wihT", dt, w_ih, w_ih_r), lambda: wih.t().contiguous(),
                          ((Cp, 2 * G4), dt, lambda out: [(out[:, :G4], w_ih.t()), (out[:, G4:], w_ih_r.t())]))
            gemm_raw(A=dG, B=wihT, C=dxn, M=R, N=Cp, K=2 * G4, a_ms=2 * G4, a_ks=1, b_ks=1, b_ns=2 * G4, c_ms=Cp, c_ns=1,
                     dtype=d)
        else:
            gemm_raw(A=dG, B=wih, C=dxn, M=R, N=Cp, K=2 * G4, a_ms=2 * G4, a_ks=1, b_ks=Cp, b_ns=1, c_ms=Cp, c_ns=1,
                     dtype=d)
        dx = torch.empty_like(x)
        pgb = _empty((nslab, Cc, 2), F32, dev)
        check(lib().dtt_gn_bwd(ptr(x), ptr(dxn), ptr(dout), ptr(gamma), ptr(gstats), ptr(dx), ptr(pgb), *geo, d,
                               stream_ptr()), "dtt_gn_bwd")
        gb = colsum(pgb, nslab * heads, Cp * 2).view(Cp, 2)
        db_f, db_r = dbias[:G4], dbias[G4:]
        return (dx, gb[:, 0].contiguous(), gb[:, 1].contiguous(), dwih[:G4], dwhh[0], db_f, db_f, dwih[G4:],
                dwhh[1], db_r, db_r, dfcw, dfcb, None, None)


# --------------------------------------------------------------------------- #
# elementwise
# --------------------------------------------------------------------------- #
class MulSkip(torch.autograd.Function):
    """out = a * skip  (decoder.py:43)."""

    @staticmethod
    def forward(ctx, a, b, slot=None):
        ctx.save_for_backward(a, b)
        ctx.slot = slot
        return binary(1, a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        ga, gb = binary(1, g, b), binary(1, g, a)
        if ctx.slot is not None and ctx.needs_input_grad[1]:
            ctx.slot.put(gb)                 # the skip's other consumer (the encoder's down-sampling conv) adds it
            gb = None
        return ga, gb, None


class InitConv(torch.autograd.Function):
    """stack([spec, phase], 1) -> 1x1 conv (encoder.py:26-27), planar fp32 in, channels-last out."""

    @staticmethod
    def forward(ctx, spec, phase, w, b, dt):
        B, Fq, T = spec.shape
        Cc = w.shape[0]
        spec = spec.contiguous().float()
        phase = phase.contiguous().float()
        out = _empty((B, Fq, T, Cc), dt, spec.device)
        w2 = w.reshape(Cc, 2).contiguous()
        check(lib().dtt_init_conv_fwd(ptr(spec), ptr(phase), ptr(w2), ptr(b), ptr(out), C.c_int64(B * Fq * T), Cc,
                                      dtype_code(dt), stream_ptr()), "dtt_init_conv_fwd")
        ctx.save_for_backward(spec, phase)
        ctx.wshape = w.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        spec, phase = ctx.saved_tensors
        dout = dout.contiguous()
        Cc = dout.shape[3]
        P = spec.numel()
        nch = int(max(1, min(1024, P // 256)))
        part = _empty((nch, Cc * 3), F32, spec.device)
        check(lib().dtt_init_conv_bwd(ptr(spec), ptr(phase), ptr(dout), ptr(part), C.c_int64(P), Cc, nch, _dt(dout),
                                      stream_ptr()), "dtt_init_conv_bwd")
        s = colsum(part, nch, Cc * 3).view(Cc, 3)
        return None, None, s[:, :2].reshape(ctx.wshape).contiguous(), s[:, 2].contiguous(), None


# --------------------------------------------------------------------------- #
# STFT / iSTFT (DFT as GEMM)
# --------------------------------------------------------------------------- #
_BASIS = {}


def _ld(n_fft):
    Fq = n_fft // 2 + 1
    imoff = (Fq + 7) // 8 * 8
    return Fq, imoff, 2 * imoff


def dft_bases(n_fft, dev):
    """fp32 forward (Hann-windowed) and inverse real-DFT bases, rows = output index."""
    key = (n_fft, str(dev))
    if key not in _BASIS:
        Fq, imoff, ld = _ld(n_fft)
        n = torch.arange(n_fft, dtype=torch.float64)
        k = torch.arange(Fq, dtype=torch.float64)
        ang = 2.0 * math.pi * torch.outer(k, n) / n_fft                       # [F][n]
        win = 0.5 - 0.5 * torch.cos(2.0 * math.pi * n / n_fft)
        fwd = torch.zeros(ld, n_fft, dtype=torch.float64)                      # [col][n]
        fwd[:Fq] = torch.cos(ang) * win
        fwd[imoff:imoff + Fq] = -torch.sin(ang) * win
        wk = torch.full((Fq,), 2.0, dtype=torch.float64)
        wk[0] = 1.0
        wk[-1] = 1.0
        inv = torch.zeros(n_fft, ld, dtype=torch.float64)                      # [n][k]
        inv[:, :Fq] = (torch.cos(ang) * wk[:, None] / n_fft).t()
        im = (-torch.sin(ang) * wk[:, None] / n_fft).t()
        im[:, 0] = 0.0
        im[:, -1] = 0.0
        inv[:, imoff:imoff + Fq] = im
        _BASIS[key] = (fwd.float().to(dev).contiguous(), inv.float().to(dev).contiguous())
    return _BASIS[key]


def stft_mag_phase(x, n_fft=1024, hop=256):
    """GPU STFT front end (replaces torchaudio Spectrogram + abs/angle in the data
    loader, base_dataset.py:154-167).  Exact-fp32 MFMA DFT GEMM.  x [B,L] -> mag, phase [B,F,T]."""
    x = x.contiguous().float()
    B, L = x.shape
    Fq, imoff, ld = _ld(n_fft)
    pad = n_fft // 2
    Lp = L + 2 * pad
    T = 1 + L // hop
    fwd, _ = dft_bases(n_fft, x.device)
    xp = _empty((B, Lp), F32, x.device)
    check(lib().dtt_reflect_pad(ptr(x), ptr(xp), B, L, pad, stream_ptr()), "dtt_reflect_pad")
    reim = _empty((B, T, ld), F32, x.device)
    gemm_raw(A=xp, B=fwd, C=reim, M=T, N=ld, K=n_fft, batch=B, a_ms=hop, a_ks=1, a_bs=Lp, b_ks=1, b_ns=n_fft,
             b_bs=0, c_ms=ld, c_ns=1, c_bs=T * ld, dtype=0)
    mag = _empty((B, Fq, T), F32, x.device)
    phase = _empty((B, Fq, T), F32, x.device)
    check(lib().dtt_stft_post(ptr(reim), ptr(mag), ptr(phase), B, Fq, T, ld, imoff, stream_ptr()), "dtt_stft_post")
    return mag, phase


class MaskISTFT(torch.autograd.Function):
    """out_conv (1x1, C -> 2S) + complex-style mask + torch.polar + rectangular iSTFT
    (decoder.py:29, model_dttnet.py:47-85).  Returns pspec, pphase [S,B,F,T], audio [S,B,L] (fp32)."""

    @staticmethod
    def forward(ctx, x, w, b, mspec, mphase, S, n_fft, hop, length):
        B, Fq, T, Cc = x.shape
        dev, dt, d = x.device, x.dtype, _dt(x)
        Fq2, imoff, ld = _ld(n_fft)
        if Fq2 != Fq:
            raise DttError(f"mask: F={Fq} does not match n_fft={n_fft}")
        mspec = mspec.contiguous().float()
        mphase = mphase.contiguous().float()
        w2 = w.reshape(2 * S, Cc).contiguous()
        pspec = _empty((S, B, Fq, T), F32, dev)
        pphase = _empty((S, B, Fq, T), F32, dev)
        # polar + iSTFT stay fp32 in every mode, as the reference's do under bf16 autocast (torch.polar /
        # torch.istft are complex64): exact-fp32 MFMA GEMM, K ~ 1032 -- under 1 % of the step
        Z = torch.zeros(S * B, T, ld, dtype=F32, device=dev)
        check(lib().dtt_mask_fwd(ptr(x), ptr(w2), ptr(b), ptr(mspec), ptr(mphase), ptr(pspec), ptr(pphase), ptr(Z),
                                 B, Fq, T, Cc, S, ld, imoff, d, stream_ptr()), "dtt_mask_fwd")
        _, inv = dft_bases(n_fft, dev)
        frames = _empty((S * B * T, n_fft), F32, dev)
        gemm_raw(A=Z, B=inv, C=frames, M=S * B * T, N=n_fft, K=ld, a_ms=ld, a_ks=1, b_ks=1, b_ns=ld, c_ms=n_fft,
                 c_ns=1, dtype=0)
        audio = _empty((S, B, length), F32, dev)
        check(lib().dtt_ola_fwd(ptr(frames), ptr(audio), S * B, T, n_fft, hop, length, stream_ptr()), "dtt_ola_fwd")
        ctx.save_for_backward(x, w2, mspec, mphase, pspec, pphase, inv)
        ctx.cfg = (S, n_fft, hop, length, w.shape)
        return pspec, pphase, audio

    @staticmethod
    def backward(ctx, dspec, dphase, daudio):
        x, w2, mspec, mphase, pspec, pphase, invt = ctx.saved_tensors
        S, n_fft, hop, length, wshape = ctx.cfg
        B, Fq, T, Cc = x.shape
        dev, dt, d = x.device, x.dtype, _dt(x)
        _, imoff, ld = _ld(n_fft)
        Lp = n_fft + hop * (T - 1)
        daudio = daudio.contiguous().float()
        gpad = _empty((S * B, Lp), F32, dev)
        check(lib().dtt_ola_bwd(ptr(daudio), ptr(gpad), S * B, T, n_fft, hop, length, 0, stream_ptr()), "dtt_ola_bwd")
        dZ = _empty((S * B, T, ld), F32, dev)
        gemm_raw(A=gpad, B=invt, C=dZ, M=T, N=ld, K=n_fft, batch=S * B, a_ms=hop, a_ks=1, a_bs=Lp, b_ks=ld, b_ns=1,
                 b_bs=0, c_ms=ld, c_ns=1, c_bs=T * ld, dtype=0)
        dx = torch.empty_like(x)
        dmask = _empty((B * Fq * T, 2 * S), dt, dev)
        check(lib().dtt_mask_bwd(ptr(w2), ptr(mspec), ptr(mphase), ptr(pspec), ptr(pphase),
                                 ptr(dspec.contiguous().float()), ptr(dphase.contiguous().float()), ptr(dZ), ptr(dx),
                                 ptr(dmask), B, Fq, T, Cc, S, ld, imoff, d, stream_ptr()), "dtt_mask_bwd")
        P = B * Fq * T
        if 2 * S <= 8 and Cc % (8 if dt == torch.bfloat16 else 4) == 0:
            nch = int(max(1, min(1024, P // 512)))
            part = _empty((nch, 2 * S * Cc), F32, dev)
            check(lib().dtt_wcolsum(ptr(dmask), ptr(x), ptr(part), C.c_int64(P), Cc, 2 * S, nch, d, stream_ptr()),
                  "dtt_wcolsum")
            dw = colsum(part, nch, 2 * S * Cc).view(2 * S, Cc)
        else:
            dw = _zeros_f32((2 * S, Cc,), dev)
            gemm_raw(A=dmask, B=x, C=dw, M=2 * S, N=Cc, K=P, splitk=_splitk(2 * S, Cc, P), a_ms=1, a_ks=2 * S,
                     b_ks=Cc, b_ns=1, c_ms=Cc, c_ns=1, dtype=d, atomic=1, out_f32=1)
        db = colsum(dmask, P, 2 * S)
        return dx, dw.view(wshape), db, None, None, None, None, None, None


# --------------------------------------------------------------------------- #
# PIT L1 loss and SI-SNR
# --------------------------------------------------------------------------- #
def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


class PitL1Loss(torch.autograd.Function):
    """DTTNetLoss under PITLossWrapper / DTTNetLossWrapper (dttnet_loss.py:28-34,
    pit_loss_wrapper.py:41-71, dttnet_loss_wrapper.py:41-62).  Permutation search on the
    device: no host sync.  Returns the per-item losses [B] and batch_permuts [B,S]."""

    @staticmethod
    def forward(ctx, pspec, pphase, paudio, fixed, *src):
        S, B = pspec.shape[0], pspec.shape[1]
        dev = pspec.device
        pspec, pphase, paudio = pspec.contiguous().float(), pphase.contiguous().float(), paudio.contiguous().float()
        src = [s.contiguous().float() for s in src]
        sspec, sphase, saudio = src[:S], src[S:2 * S], src[2 * S:]
        FT = pspec.shape[2] * pspec.shape[3]
        L = paudio.shape[2]
        nc_sp = int(max(1, min(64, FT // 4096)))
        nc_au = int(max(1, min(64, L // 4096)))
        p_sp = _empty((B, nc_sp, S * S), F32, dev)
        p_ph = _empty((B, nc_sp, S * S), F32, dev)
        p_au = _empty((B, nc_au, S * S), F32, dev)
        st = stream_ptr()
        check(lib().dtt_l1_pairs(ptr(pspec), _ptr_array(sspec), ptr(p_sp), S, B, C.c_int64(FT), nc_sp, st), "dtt_l1_pairs")
        check(lib().dtt_l1_pairs(ptr(pphase), _ptr_array(sphase), ptr(p_ph), S, B, C.c_int64(FT), nc_sp, st), "dtt_l1_pairs")
        check(lib().dtt_l1_pairs(ptr(paudio), _ptr_array(saudio), ptr(p_au), S, B, C.c_int64(L), nc_au, st), "dtt_l1_pairs")
        item = _empty((B,), F32, dev)
        permuts = _empty((B, S), torch.int64, dev)
        loss = _empty((), F32, dev)
        check(lib().dtt_pit_select(ptr(p_sp), ptr(p_ph), ptr(p_au), nc_sp, nc_au, C.c_int64(FT), C.c_int64(L), S, B,
                                   1 if fixed else 0, ptr(item), ptr(permuts), ptr(loss), st), "dtt_pit_select")
        ctx.save_for_backward(pspec, pphase, paudio, permuts, *src)
        ctx.mark_non_differentiable(permuts)
        return item, permuts

    @staticmethod
    def backward(ctx, gloss, _gp):
        pspec, pphase, paudio, permuts, *src = ctx.saved_tensors
        S, B = pspec.shape[0], pspec.shape[1]
        FT = pspec.shape[2] * pspec.shape[3]
        L = paudio.shape[2]
        gloss = gloss.contiguous().float().reshape(B)          # per-item upstream gradient
        st = stream_ptr()
        outs = []
        for pred, srcs, n in ((pspec, src[:S], FT), (pphase, src[S:2 * S], FT), (paudio, src[2 * S:], L)):
            dp = torch.empty_like(pred)
            check(lib().dtt_l1_bwd(ptr(pred), _ptr_array(srcs), ptr(permuts), ptr(gloss), C.c_float(1.0 / n),
                                   ptr(dp), S, B, C.c_int64(n), st), "dtt_l1_bwd")
            outs.append(dp)
        return (outs[0], outs[1], outs[2], None) + (None,) * len(src)


def si_snr_matrix(est, sources, eps=1e-5):
    """est [E,B,L] fp32, sources: list of S [B,L] -> SI-SNR [E,S,B] (si_snr.py:26-34)."""
    est = est.contiguous().float()
    sources = [s.contiguous().float() for s in sources]
    E, B, L = est.shape
    S = len(sources)
    nch = int(max(1, min(32, L // 2048)))
    dev = est.device
    part = _empty((E * S * B * nch * 2,), F32, dev)
    alpha = _empty((E * S * B,), F32, dev)
    out = _empty((E, S, B), F32, dev)
    check(lib().dtt_sisnr(ptr(est), _ptr_array(sources), ptr(part), ptr(alpha), ptr(out), E, S, B, C.c_int64(L), nch,
                          C.c_float(eps), stream_ptr()), "dtt_sisnr")
    return out


# --------------------------------------------------------------------------- #
# 1x1 conv (+ InstanceNorm + GELU), FiLM, token mean / broadcast concat
# --------------------------------------------------------------------------- #
def _linear_fwd(x2d, w, b, out2d, res=None):
    """out[r][n] = x[r][:] . w[n][:] + b[n] (+ res); w is [N][K] in x's dtype."""
    R, K = x2d.shape
    N = w.shape[0]
    kw = dict(A=x2d, B=w, C=out2d, M=R, N=N, K=K, a_ms=K, a_ks=1, b_ks=1, b_ns=K, c_ms=N, c_ns=1, dtype=_dt(x2d))
    if b is not None:
        kw.update(bias=b, bias_mode=1)
    if res is not None:
        kw.update(R=res, r_ms=N, r_ns=1)
    gemm_raw(**kw)


def _linear_bwd(x2d, w, dy2d, need_dx=True, on=True):
    R, K = x2d.shape
    N = w.shape[0]
    d = _dt(x2d)
    dx = None
    if need_dx:
        dx = torch.empty_like(x2d)
        gemm_raw(A=dy2d, B=w, C=dx, M=R, N=K, K=N, a_ms=N, a_ks=1, b_ks=K, b_ns=1, c_ms=K, c_ns=1, dtype=d)
    dw = _zeros_f32((N, K,), x2d.device)
    with _lane(x2d.device, dy2d, x2d, on=on):
        gemm_raw(A=dy2d, B=x2d, C=dw, M=N, N=K, K=R, splitk=_splitk(N, K, R), a_ms=1, a_ks=N, b_ks=K,
                 b_ns=1, c_ms=K, c_ns=1, dtype=d, atomic=1, out_f32=1)
    return dx, dw


class Conv1x1(torch.autograd.Function):
    """1x1 conv on channels-last (temporal_alignment.py:9, audio_projection)."""

    @staticmethod
    def forward(ctx, x, w, b):
        Co, Ci = w.shape[0], w.shape[1]
        wp = packed(_wkey("c1", x.dtype, w), lambda: w.reshape(Co, Ci).to(x.dtype).contiguous(),
                    ((Co, Ci), x.dtype, lambda out: [(out, w.reshape(Co, Ci))]))
        y = _empty(x.shape[:-1] + (Co,), x.dtype, x.device)
        _linear_fwd(x.reshape(-1, Ci), wp, b, y.view(-1, Co))
        ctx.save_for_backward(x, wp)
        ctx.wshape, ctx.lane_params = w.shape, (w,)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wp = ctx.saved_tensors
        dy = dy.contiguous()
        Co, Ci = wp.shape
        dx, dw = _linear_bwd(x.reshape(-1, Ci), wp, dy.view(-1, Co), ctx.needs_input_grad[0], on=_lane_ok(*ctx.lane_params))
        db = colsum(dy, dy.numel() // Co, Co)
        return (dx.view_as(x) if dx is not None else None), dw.view(ctx.wshape), db


class Conv1x1NormGelu(torch.autograd.Function):
    """channel_compression: 1x1 conv -> InstanceNorm2d -> GELU (model_dttnet_vision.py:73-77)."""

    @staticmethod
    def forward(ctx, x, w, b):
        B, Fq, T, Ci = x.shape
        Co = w.shape[0]
        wp = packed(_wkey("c1", x.dtype, w), lambda: w.reshape(Co, Ci).to(x.dtype).contiguous(),
                    ((Co, Ci), x.dtype, lambda out: [(out, w.reshape(Co, Ci))]))
        y = _empty((B, Fq, T, Co), x.dtype, x.device)
        _linear_fwd(x.view(-1, Ci), wp, b, y.view(-1, Co))
        stats = in_stats(y)
        z = in_gelu_fwd(y, stats, Fq, T)
        ctx.save_for_backward(x, wp, y, stats)
        ctx.wshape, ctx.lane_params = w.shape, (w,)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, wp, y, stats = ctx.saved_tensors
        Co, Ci = wp.shape
        dy = in_gelu_bwd(y, dz.contiguous(), stats)
        dx, dw = _linear_bwd(x.view(-1, Ci), wp, dy.view(-1, Co), ctx.needs_input_grad[0], on=_lane_ok(*ctx.lane_params))
        db = _zeros_f32((Co,), x.device)       # bias ahead of InstanceNorm: exactly zero
        return (dx.view_as(x) if dx is not None else None), dw.view(ctx.wshape), db


class LerpTime(torch.autograd.Function):
    """F.interpolate(x.transpose(-1, -2), size=T, mode="linear").transpose(-1, -2) for x fp32 [B, T_in, D]
    (block_fusion.py:18-19; temporal_alignment.py:19-22, where the bilinear resize of a tensor constant over F is this
    interpolation over time): ATen's kernel walks batch x channel serially per output frame -- 4.3 ms forward and
    9.7 ms backward (atomics) at [64, 512, 50 -> 16] -- these are one gather pass each way."""

    @staticmethod
    def forward(ctx, x, T):
        x = x.contiguous().float()
        B, Tin, D = x.shape
        out = _empty((B, T, D), F32, x.device)
        check(lib().dtt_lerp_time_fwd(ptr(x), ptr(out), B, Tin, T, D, stream_ptr()), "dtt_lerp_time_fwd")
        ctx.dims = (B, Tin, T, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, Tin, T, D = ctx.dims
        dout = dout.contiguous().float()
        dx = _empty((B, Tin, D), F32, dout.device)
        check(lib().dtt_lerp_time_bwd(ptr(dout), ptr(dx), B, Tin, T, D, stream_ptr()), "dtt_lerp_time_bwd")
        return dx, None


class Film(torch.autograd.Function):
    """(1 + gamma_c + gamma_f) * x + beta_c + beta_f  (block_fusion.py:30-37).
    gbc [B,T,2C] = (gamma_c | beta_c), gbf [B,T,2F] = (gamma_f | beta_f), fp32."""

    @staticmethod
    def forward(ctx, x, gbc, gbf):
        B, Fq, T, Cc = x.shape
        gbc, gbf = gbc.contiguous().float(), gbf.contiguous().float()
        y = torch.empty_like(x)
        check(lib().dtt_film_fwd(ptr(x), ptr(gbc), ptr(gbf), ptr(y), B, Fq, T, Cc, _dt(x), stream_ptr()), "dtt_film_fwd")
        ctx.save_for_backward(x, gbc, gbf)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gbc, gbf = ctx.saved_tensors
        B, Fq, T, Cc = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgbc, dgbf = torch.empty_like(gbc), torch.empty_like(gbf)
        check(lib().dtt_film_bwd(ptr(x), ptr(dy), ptr(gbc), ptr(gbf), ptr(dx), ptr(dgbc), ptr(dgbf), B, Fq, T, Cc,
                                 _dt(x), stream_ptr()), "dtt_film_bwd")
        return dx, dgbc, dgbf


class FreqMeanTokens(torch.autograd.Function):
    """tokens[b,t,c] = mean_f x[b,f,t,c] (cross_attention_positional.py:55) -> fp32.
    With a GradSlot shared with BroadcastConcat (x's other consumer, whose backward runs first: the token network
    hangs off its enh gradient) the two partial gradients of x are formed by ONE pass, here: dx = dout[..., :C] +
    dtok / F.  Without a slot: the plain broadcast, summed by autograd."""

    @staticmethod
    def forward(ctx, x, slot=None):
        B, Fq, T, Cc = x.shape
        tok = _empty((B, T, Cc), F32, x.device)
        check(lib().dtt_fmean(ptr(x), ptr(tok), B, Fq, T, Cc, _dt(x), stream_ptr()), "dtt_fmean")
        ctx.shape = x.shape
        ctx.dt = x.dtype
        ctx.slot = slot
        return tok

    @staticmethod
    def backward(ctx, dtok):
        B, Fq, T, Cc = ctx.shape
        dout = _slot_take(ctx.slot, "bcat")
        if dout is not None:
            dtok = dtok.contiguous().float()
            dx = _empty((B, Fq, T, Cc), dout.dtype, dout.device)
            check(lib().dtt_bcat_bwd(ptr(dout), ptr(dtok), ptr(dx), None, B, Fq, T, Cc, _dt(dout), stream_ptr()),
                  "dtt_bcat_bwd")
            return dx, None
        return (dtok / Fq).to(ctx.dt).unsqueeze(1).expand(B, Fq, T, Cc).contiguous(), None


class BroadcastConcat(torch.autograd.Function):
    """cat([x, enh[b,t,c] expanded over F], channel) (cross_attention_positional.py:77-78,
    temporal_alignment.py:17-24)."""

    @staticmethod
    def forward(ctx, x, enh, slot=None):
        B, Fq, T, Cc = x.shape
        enh = enh.contiguous().float()
        out = _empty((B, Fq, T, 2 * Cc), x.dtype, x.device)
        check(lib().dtt_bcat_fwd(ptr(x), ptr(enh), ptr(out), B, Fq, T, Cc, _dt(x), stream_ptr()), "dtt_bcat_fwd")
        ctx.slot = slot
        return out

    @staticmethod
    def backward(ctx, dout):
        B, Fq, T, C2 = dout.shape
        Cc = C2 // 2
        dout = dout.contiguous()
        denh = _empty((B, T, Cc), F32, dout.device)
        if ctx.slot is not None and ctx.needs_input_grad[0]:
            # x's half waits for the token mean's gradient (FreqMeanTokens.backward forms dx in one pass)
            check(lib().dtt_bcat_bwd(ptr(dout), None, None, ptr(denh), B, Fq, T, Cc, _dt(dout), stream_ptr()),
                  "dtt_bcat_bwd")
            ctx.slot.put(dout, "bcat")
            return None, denh, None
        dx = _empty((B, Fq, T, Cc), dout.dtype, dout.device)
        check(lib().dtt_bcat_bwd(ptr(dout), None, ptr(dx), ptr(denh), B, Fq, T, Cc, _dt(dout), stream_ptr()),
              "dtt_bcat_bwd")
        return dx, denh, None


# --------------------------------------------------------------------------- #
# BatchNorm (+ residual) + activation of the lip encoders (channels-last)
# --------------------------------------------------------------------------- #
class BatchNormAct(torch.autograd.Function):
    """act(batch_norm(x) (+ res)) on an NCHW tensor held in channels-last memory (what the convolutions of the
    lip encoder produce under bf16 autocast).  VideoEncoders/resnet.py:36-81: bn1 -> swish, bn2 -> + residual ->
    swish, downsample bn (act 0).  Batch statistics + running-buffer update in training mode, running statistics in
    eval mode; nothing but x (and res) is saved: backward recomputes the activation input."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, rmean, rvar, training, momentum, eps, act, slot=None):
        ctx.slot = slot
        if x.dim() != 4:
            raise DttError("BatchNormAct expects an NCHW tensor")
        x = x.contiguous(memory_format=torch.channels_last)
        if res is not None:
            res = res.to(x.dtype).contiguous(memory_format=torch.channels_last)
        Nb, Cc, Hh, Ww = x.shape
        P = Nb * Hh * Ww
        d, dev = _dt(x), x.device
        stats = _empty((Cc, 2), F32, dev)
        if training:
            part = _empty((lib().dtt_bn_chunks(P), Cc, 2), F32, dev)
            check(lib().dtt_bn_stats(ptr(x), ptr(part), ptr(stats), P, Cc, C.c_float(eps), d, stream_ptr()),
                  "dtt_bn_stats")
            if rmean is not None:
                check(lib().dtt_bn_running(ptr(stats), ptr(rmean), ptr(rvar), Cc, C.c_int64(P), C.c_float(momentum),
                                           C.c_float(eps), stream_ptr()), "dtt_bn_running")
        else:
            check(lib().dtt_bn_frozen_stats(ptr(rmean), ptr(rvar), ptr(stats), Cc, C.c_float(eps), stream_ptr()),
                  "dtt_bn_frozen_stats")
        z = torch.empty_like(x)                                 # channels-last, as x
        check(lib().dtt_bn_act_fwd(ptr(x), ptr(res), ptr(stats), ptr(gamma), ptr(beta), ptr(z), P, Cc, act, d,
                                   stream_ptr()), "dtt_bn_act_fwd")
        ctx.save_for_backward(x, res, stats, gamma, beta)
        ctx.cfg = (act, bool(training))
        return z

    @staticmethod
    def backward(ctx, dz):
        x, res, stats, gamma, beta = ctx.saved_tensors
        act, training = ctx.cfg
        Nb, Cc, Hh, Ww = x.shape
        P = Nb * Hh * Ww
        d, dev = _dt(x), x.device
        dz = dz.to(x.dtype).contiguous(memory_format=torch.channels_last)
        nch = lib().dtt_bn_chunks(P)
        part = _empty((nch, Cc, 2), F32, dev)
        sums = _empty((Cc, 2), F32, dev)
        dgb = _empty((Cc, 2), F32, dev)
        dx = torch.empty_like(x)
        need_res = res is not None and ctx.needs_input_grad[1]
        dres = torch.empty_like(x) if need_res else None
        check(lib().dtt_bn_act_bwd(ptr(x), ptr(res), ptr(dz), ptr(stats), ptr(gamma), ptr(beta), ptr(part), ptr(sums),
                                   ptr(dgb), ptr(dx), ptr(dres), P, Cc, act, 1 if training else 0, d, stream_ptr()),
              "dtt_bn_act_bwd")
        if ctx.slot is not None and dres is not None:
            ctx.slot.put(dres)               # the residual's other consumer (the block's conv1) adds it in its dgrad
            dres = None
        return dx, dres, dgb[:, 0].contiguous(), dgb[:, 1].contiguous(), None, None, None, None, None, None, None


def _bn_momentum(bn):
    """torch's momentum=None means a CUMULATIVE average (factor 1 / num_batches_tracked, a device counter): the
    reference's lip encoders never use it (resnet.py:36-81: default 0.1), so it is refused rather than silently
    replaced by 0.1."""
    if bn.momentum is None:
        if bn.training and bn.track_running_stats:
            raise DttError("BatchNorm with momentum=None (cumulative moving average) is not supported by the fused "
                           "BatchNorm kernels; use a numeric momentum")
        return 0.0
    return float(bn.momentum)


def batch_norm_act(x, bn, act, res=None, slot=None):
    """``bn``: an nn.BatchNorm2d (parameter / buffer container; its train()/eval() mode is honoured).  slot: GradSlot
    shared with the residual's other consumer."""
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    mom = _bn_momentum(bn)
    use_batch = bn.training or not bn.track_running_stats
    return BatchNormAct.apply(x, res, bn.weight, bn.bias, bn.running_mean, bn.running_var, use_batch, mom, bn.eps, act, slot)


_POOL_BWD_FUSED = os.environ.get("DTT_POOL_BWD_FUSED", "0") == "1"


class BatchNormActPool(torch.autograd.Function):
    """max_pool2d(act(batch_norm(x)), 3, 2, 1) in one pass each way (the lip front end, resnet_video_encoder.py:38-44):
    the pre-pool activation tensor (400 MB at 1600 frames) and its gradient are never written.  x: NCHW in
    channels-last memory; saves x, the statistics and one argmax byte per pooled element."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, training, momentum, eps, act):
        x = x.contiguous(memory_format=torch.channels_last)
        Nb, Cc, Hh, Ww = x.shape
        P = Nb * Hh * Ww
        d, dev = _dt(x), x.device
        stats = _empty((Cc, 2), F32, dev)
        if training:
            part = _empty((lib().dtt_bn_chunks(P), Cc, 2), F32, dev)
            check(lib().dtt_bn_stats(ptr(x), ptr(part), ptr(stats), P, Cc, C.c_float(eps), d, stream_ptr()),
                  "dtt_bn_stats")
            if rmean is not None:
                check(lib().dtt_bn_running(ptr(stats), ptr(rmean), ptr(rvar), Cc, C.c_int64(P), C.c_float(momentum),
                                           C.c_float(eps), stream_ptr()), "dtt_bn_running")
        else:
            check(lib().dtt_bn_frozen_stats(ptr(rmean), ptr(rvar), ptr(stats), Cc, C.c_float(eps), stream_ptr()),
                  "dtt_bn_frozen_stats")
        Ho, Wo = (Hh - 1) // 2 + 1, (Ww - 1) // 2 + 1
        y = torch.empty((Nb, Cc, Ho, Wo), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        idx = torch.empty((Nb, Ho, Wo, Cc), dtype=torch.uint8, device=dev)
        check(lib().dtt_bn_act_pool_fwd(ptr(x), ptr(stats), ptr(gamma), ptr(beta), ptr(y), ptr(idx), Nb, Hh, Ww, Cc, act, d,
                                        stream_ptr()), "dtt_bn_act_pool_fwd")
        ctx.save_for_backward(x, stats, gamma, beta, idx)
        ctx.cfg = (act, bool(training))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, gamma, beta, idx = ctx.saved_tensors
        act, training = ctx.cfg
        Nb, Cc, Hh, Ww = x.shape
        P = Nb * Hh * Ww
        d, dev = _dt(x), x.device
        dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        nch = lib().dtt_bn_chunks(P)
        part = _empty((nch, Cc, 2), F32, dev)
        sums = _empty((Cc, 2), F32, dev)
        dgb = _empty((Cc, 2), F32, dev)
        dx = torch.empty_like(x)
        if _POOL_BWD_FUSED:
            check(lib().dtt_bn_act_pool_bwd(ptr(x), ptr(dy), ptr(idx), ptr(stats), ptr(gamma), ptr(beta), ptr(part),
                                            ptr(sums), ptr(dgb), ptr(dx), Nb, Hh, Ww, Cc, act, 1 if training else 0, d,
                                            stream_ptr()), "dtt_bn_act_pool_bwd")
        else:
            # measured (cfg4, same box): gathering the pooled gradient inside both BatchNorm passes (4 candidate
            # windows per pixel, 190-210 registers) is 0.3 ms slower than scattering it once and running the plain
            # passes over the 400 MB tensor; the fused FORWARD is 0.08 ms faster than the pair
            dz = torch.empty_like(x)
            check(lib().dtt_maxpool3x3s2_bwd(ptr(dy), ptr(idx), ptr(dz), Nb, Hh, Ww, Cc, d, stream_ptr()),
                  "dtt_maxpool3x3s2_bwd")
            check(lib().dtt_bn_act_bwd(ptr(x), None, ptr(dz), ptr(stats), ptr(gamma), ptr(beta), ptr(part), ptr(sums),
                                       ptr(dgb), ptr(dx), None, P, Cc, act, 1 if training else 0, d, stream_ptr()),
                  "dtt_bn_act_bwd")
        return dx, dgb[:, 0].contiguous(), dgb[:, 1].contiguous(), None, None, None, None, None, None


def batch_norm_act_pool(x, bn, act):
    """batch_norm_act followed by MaxPool3x3s2, fused (``bn`` as in batch_norm_act)."""
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    mom = _bn_momentum(bn)
    use_batch = bn.training or not bn.track_running_stats
    return BatchNormActPool.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, use_batch, mom, bn.eps, act)


class MaxPool3x3s2(torch.autograd.Function):
    """max_pool2d(x, 3, 2, 1) on an NCHW tensor in channels-last memory (the per-frame form of the lip encoders'
    MaxPool3d((1,3,3),(1,2,2),(0,1,1)))."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous(memory_format=torch.channels_last)
        Nb, Cc, Hh, Ww = x.shape
        Ho, Wo = (Hh - 1) // 2 + 1, (Ww - 1) // 2 + 1
        y = torch.empty((Nb, Cc, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((Nb, Ho, Wo, Cc), dtype=torch.uint8, device=x.device)
        check(lib().dtt_maxpool3x3s2_fwd(ptr(x), ptr(y), ptr(idx), Nb, Hh, Ww, Cc, _dt(x), stream_ptr()),
              "dtt_maxpool3x3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (Nb, Cc, Hh, Ww)
        return y

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        Nb, Cc, Hh, Ww = ctx.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty((Nb, Cc, Hh, Ww), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        check(lib().dtt_maxpool3x3s2_bwd(ptr(dy), ptr(idx), ptr(dx), Nb, Hh, Ww, Cc, _dt(dy), stream_ptr()),
              "dtt_maxpool3x3s2_bwd")
        return dx


# --------------------------------------------------------------------------- #
# token network of the cross-attention fusion (cross_attention_positional.py:55-75)
# --------------------------------------------------------------------------- #
class Linear(torch.autograd.Function):
    """y = x W^T + b through dtt_gemm (exact-fp32 MFMA for fp32 tokens), optional fused sigmoid (the audio gate,
    cross_attention_positional.py:46-49); weight / bias gradients by split-K GEMM and deterministic column sums."""

    @staticmethod
    def forward(ctx, x, w, b, sigmoid):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        wt = packed(_wkey("lin", x2.dtype, w) + (w.storage_offset(), w.shape[0]), lambda: w.to(x2.dtype).contiguous(),
                    (tuple(w.shape), x2.dtype, lambda out: [(out, w)]))
        y = _empty((x2.shape[0], wt.shape[0]), x2.dtype, x2.device)
        kw = dict(A=x2, B=wt, C=y, M=x2.shape[0], N=wt.shape[0], K=x2.shape[1], a_ms=x2.shape[1], a_ks=1, b_ks=1,
                  b_ns=x2.shape[1], c_ms=wt.shape[0], c_ns=1, dtype=_dt(x2), act=2 if sigmoid else 0)
        if b is not None:
            kw.update(bias=b, bias_mode=1)
        gemm_raw(**kw)
        ctx.save_for_backward(x2, wt, y if sigmoid else None)
        ctx.has_bias = b is not None
        ctx.xshape, ctx.lane_params = x.shape, (w,)
        return y.view(x.shape[:-1] + (wt.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x2, wt, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, wt.shape[0]).to(x2.dtype)
        if y is not None:
            dy2 = dy2 * (y * (1.0 - y))
        dy2 = dy2.contiguous()
        # (a weight that is a slice of a parameter -- the q / k / v thirds of in_proj_weight -- has a backward node that
        # reads dw on the launch stream: no side stream for it)
        dx, dw = _linear_bwd(x2, wt, dy2, ctx.needs_input_grad[0], on=_lane_ok(*ctx.lane_params))
        db = colsum(dy2, dy2.shape[0], dy2.shape[1]) if ctx.has_bias else None
        return (dx.view(ctx.xshape) if dx is not None else None), dw, db, None


class CrossAttnCore(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh)) v per head on exact-fp32 MFMA tiles (attn.hip): the core of the reference's
    nn.MultiheadAttention call (cross_attention_positional.py:66-69) behind its input projections.
    q [B,Tq,C], k, v [B,Tk,C] fp32 -> [B,Tq,C]."""

    @staticmethod
    def forward(ctx, q, k, v, heads):
        q, k, v = q.contiguous().float(), k.contiguous().float(), v.contiguous().float()
        B, Tq, Cc = q.shape
        Tk = k.shape[1]
        scale = 1.0 / math.sqrt(Cc // heads)
        out = _empty((B, Tq, Cc), F32, q.device)
        prob = _empty((B, heads, Tq, Tk), F32, q.device)
        check(lib().dtt_xattn_fwd(ptr(q), ptr(k), ptr(v), ptr(out), ptr(prob), B, Tq, Tk, Cc, heads, C.c_float(scale),
                                  stream_ptr()), "dtt_xattn_fwd")
        ctx.save_for_backward(q, k, v, prob)
        ctx.cfg = (heads, scale)
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v, prob = ctx.saved_tensors
        heads, scale = ctx.cfg
        B, Tq, Cc = q.shape
        Tk = k.shape[1]
        do = do.contiguous().float()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        check(lib().dtt_xattn_bwd(ptr(q), ptr(k), ptr(v), ptr(prob), ptr(do), ptr(dq), ptr(dk), ptr(dv), B, Tq, Tk, Cc,
                                  heads, C.c_float(scale), stream_ptr()), "dtt_xattn_bwd")
        return dq, dk, dv, None
