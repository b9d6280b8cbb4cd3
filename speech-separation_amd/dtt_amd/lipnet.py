"""The FiLM model's frozen lip-reading front end (Conv3d stem + ShuffleNetV2 1.0x trunk) on the HIP kernels, forward
only -- the encoder's parameters are frozen (DTTNetPlus/modules/video_encoder.py:38-39) but its BatchNorm layers
follow train()/eval(): batch statistics and running-buffer updates while the model trains.

Reference: src/model/lipreading/model.py:164-174 (frontend3D -> per-frame trunk), src/model/lipreading/models/
shufflenetv2.py:42-104 (units), :26-40 (channel_shuffle), :151-152 (conv_last + globalpool).

How a unit runs (lipunit.hip has the layout and the kernels):
  1x1 conv            dtt_gemm on the channels-last rows (the second half of a stride-1 unit's input is a column
                      range of those rows: no split copy)
  BN + ReLU behind it applied by the depthwise kernel to the taps it loads (dtt_dwconv3x3_bnrelu)
  BN behind the depthwise conv   folded into the next 1x1 conv's weights + a bias (dtt_pw_fold_bn)
  BN + ReLU + concat + channel_shuffle   one pass (dtt_bn_relu_shuffle)
  conv_last's BN + ReLU + AvgPool2d(3)   one pass (dtt_bn_relu_avgpool)
Batch statistics and the running-buffer update: dtt_bn_stats_running over the raw conv output (bn_act.hip).
The weights are frozen, so their channel-padded copies are built once per (device, dtype) and rebuilt only when a
parameter's version counter moves (load_state_dict).
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import DttError, check, dtype_code, gemm_raw, lib, ptr, stream_ptr

F32 = torch.float32


def _r8(n):
    return (n + 7) // 8 * 8


class _Layout:
    """Columns of a channels-last activation: dense (C channels + zero columns up to a multiple of 8) or two halves of
    h = C / 2 channels, each padded to hp.  segs: (first column, first logical channel, count)."""

    def __init__(self, channels, halves):
        self.C = channels
        if halves:
            self.h = channels // 2
            self.hp = _r8(self.h)
            self.Cp = 2 * self.hp
            self.segs = [(0, 0, self.h), (self.hp, self.h, self.h)]
        else:
            self.Cp = _r8(channels)
            self.segs = [(0, 0, channels)]

    def spread(self, v):
        """logical [..., C] -> physical [..., Cp] (zeros in the padding columns)"""
        out = torch.zeros(v.shape[:-1] + (self.Cp,), dtype=F32, device=v.device)
        for po, lo, n in self.segs:
            out[..., po:po + n] = v[..., lo:lo + n]
        return out


class _BN:
    def __init__(self, bn, lay):
        self.bn, self.lay = bn, lay
        self.gamma = lay.spread(bn.weight.detach().float()).contiguous()
        self.beta = lay.spread(bn.bias.detach().float()).contiguous()


def _pw_weight(conv, lay_in, rows):
    """[Co, Ci, 1, 1] -> fp32 [rows >= Co][lay_in.Cp]: input channels at their columns, zero rows / columns as padding"""
    w = lay_in.spread(conv.weight.detach().float().reshape(conv.out_channels, conv.in_channels))
    out = torch.zeros((rows, lay_in.Cp), dtype=F32, device=w.device)
    out[:conv.out_channels] = w
    return out


def _dw_weight(conv, lay):
    """[C, 1, 3, 3] -> fp32 [9 taps][Cp]"""
    return lay.spread(conv.weight.detach().float().reshape(conv.out_channels, 9).t().contiguous()).contiguous()


def _check_unit(u, inp):
    b2 = u.banch2
    ok = (len(b2) == 8 and b2[0].kernel_size == (1, 1) and b2[3].kernel_size == (3, 3) and b2[3].groups == b2[3].in_channels
          and b2[3].padding == (1, 1) and b2[5].kernel_size == (1, 1) and all(c.bias is None for c in (b2[0], b2[3], b2[5])))
    if u.down:
        b1 = u.banch1
        ok = ok and (len(b1) == 5 and b1[0].kernel_size == (3, 3) and b1[0].groups == inp and b1[0].stride == (2, 2)
                     and b1[2].kernel_size == (1, 1) and b2[3].stride == (2, 2) and b2[0].in_channels == inp)
    else:
        ok = ok and b2[3].stride == (1, 1) and b2[0].in_channels * 2 == inp
    if not ok:
        raise DttError("lipnet: not a ShuffleNetV2 unit of the reference's form")


def _build(enc, dt):
    """Channel-padded fp32 / dtype copies of the frozen weights, per unit."""
    units, conv5 = enc.trunk[0], enc.trunk[1]
    lay = _Layout(units[0].banch2[0].in_channels, False)            # the stem's output: dense
    packs = []
    with torch.no_grad():
        for u in units:
            _check_unit(u, lay.C)
            b2 = u.banch2
            h = b2[5].out_channels
            mid = _Layout(h, False)                                 # branch tensors: dense, h channels + padding
            out = _Layout(2 * h, True)
            if h % 2:
                raise DttError("lipnet: odd branch width")
            pk = dict(down=u.down, h=h, hp=mid.Cp, lay_in=lay)
            if u.down:
                b1 = u.banch1
                pk.update(wd1=_dw_weight(b1[0], lay), bn1a=_BN(b1[1], lay), w1=_pw_weight(b1[2], lay, mid.Cp), bn1b=_BN(b1[3], mid),
                          w2a=_pw_weight(b2[0], lay, mid.Cp).to(dt))
            else:
                if lay.Cp != 2 * mid.Cp:
                    raise DttError("lipnet: a stride-1 unit must keep the channel count")
                pk.update(w2a=_pw_weight(b2[0], mid, mid.Cp).to(dt))
            pk.update(bn2a=_BN(b2[1], mid), wd2=_dw_weight(b2[3], mid), bn2b=_BN(b2[4], mid), w2c=_pw_weight(b2[5], mid, mid.Cp),
                      bn2c=_BN(b2[6], mid))
            packs.append(pk)
            lay = out
        c5 = conv5[0]
        if not (c5.kernel_size == (1, 1) and c5.bias is None and c5.in_channels == lay.C and c5.out_channels % 8 == 0):
            raise DttError("lipnet: unexpected conv_last")
        last = dict(w=_pw_weight(c5, lay, c5.out_channels).to(dt), bn=_BN(conv5[1], _Layout(c5.out_channels, False)), lay_in=lay)
    return packs, last


def _packs(enc, dt, dev):
    ver = sum(p._version for p in enc.parameters())
    key = (str(dev), dt)
    cache = enc.__dict__.setdefault("_lip_packs", {})
    hit = cache.get(key)
    if hit is None or hit[0] != ver:
        hit = (ver, _build(enc, dt))
        cache[key] = hit
    return hit[1]


class _Run:
    """One forward: the BatchNorm bookkeeping (batch / running statistics, buffer updates) and the launches."""

    def __init__(self, dt, dev):
        self.dt, self.dev, self.d = dt, dev, dtype_code(dt)
        self.tracked = []

    def stats(self, y, P, b):
        bn, Cp = b.bn, b.lay.Cp
        if bn.training or not bn.track_running_stats:
            stats = torch.empty((Cp, 2), dtype=F32, device=self.dev)
            part = torch.empty((lib().dtt_bn_chunks(P), Cp, 2), dtype=F32, device=self.dev)
            if bn.training and bn.track_running_stats and bn.running_mean is not None:
                if bn.momentum is None:
                    raise DttError("lipnet: BatchNorm momentum=None (cumulative average) is not supported")
                h, hp = (b.lay.h, b.lay.hp) if len(b.lay.segs) == 2 else (Cp, Cp)
                check(lib().dtt_bn_stats_running(ptr(y), ptr(part), ptr(stats), P, Cp, C.c_float(bn.eps), self.d,
                                                 ptr(bn.running_mean), ptr(bn.running_var), C.c_float(bn.momentum), h, hp,
                                                 stream_ptr()), "dtt_bn_stats_running")
                if bn.num_batches_tracked is not None:
                    self.tracked.append(bn.num_batches_tracked)
            else:
                check(lib().dtt_bn_stats(ptr(y), ptr(part), ptr(stats), P, Cp, C.c_float(bn.eps), self.d, stream_ptr()),
                      "dtt_bn_stats")
            return stats
        stats = torch.zeros((Cp, 2), dtype=F32, device=self.dev)
        for po, lo, n in b.lay.segs:
            check(lib().dtt_bn_frozen_stats(ptr(bn.running_mean[lo:]), ptr(bn.running_var[lo:]), ptr(stats[po:]), n,
                                            C.c_float(bn.eps), stream_ptr()), "dtt_bn_frozen_stats")
        return stats

    def pw(self, a, lda, K, w, bias, P):
        """rows of `a` (row stride lda, K columns) x w[N][K]^T (+ bias) -> [P, N]"""
        N = w.shape[0]
        y = torch.empty((P, N), dtype=self.dt, device=self.dev)
        kw = dict(A=a, B=w, C=y, M=P, N=N, K=K, a_ms=lda, a_ks=1, b_ks=1, b_ns=K, c_ms=N, c_ns=1, dtype=self.d)
        if bias is not None:
            kw.update(bias=bias, bias_mode=1)
        gemm_raw(**kw)
        return y

    def dw(self, x, NF, H, W, Cp, w9c, pre, stride):
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty((NF * Ho * Wo, Cp), dtype=self.dt, device=self.dev)
        st, g, b = (ptr(pre[0]), ptr(pre[1].gamma), ptr(pre[1].beta)) if pre is not None else (None, None, None)
        check(lib().dtt_dwconv3x3_bnrelu(ptr(x), ptr(w9c), st, g, b, ptr(y), NF, H, W, Cp, stride, self.d, stream_ptr()),
              "dtt_dwconv3x3_bnrelu")
        return y, Ho, Wo

    def fold(self, w, stats, b):
        wout = torch.empty(w.shape, dtype=self.dt, device=self.dev)
        bias = torch.empty((w.shape[0],), dtype=F32, device=self.dev)
        check(lib().dtt_pw_fold_bn(ptr(w), ptr(stats), ptr(b.gamma), ptr(b.beta), ptr(wout), ptr(bias), w.shape[0], w.shape[1],
                                   self.d, stream_ptr()), "dtt_pw_fold_bn")
        return wout, bias

    def shuffle(self, a, lda, sa, b, ldb, sb, P, h, hp):
        out = torch.empty((P, 2 * hp), dtype=self.dt, device=self.dev)
        pa = (ptr(sa[0]), ptr(sa[1].gamma), ptr(sa[1].beta)) if sa is not None else (None, None, None)
        pb = (ptr(sb[0]), ptr(sb[1].gamma), ptr(sb[1].beta)) if sb is not None else (None, None, None)
        check(lib().dtt_bn_relu_shuffle(ptr(a), lda, *pa, ptr(b), ldb, *pb, ptr(out), C.c_int64(P), h, hp, self.d,
                                        stream_ptr()), "dtt_bn_relu_shuffle")
        return out

    def unit(self, pk, x, NF, H, W):
        """x [NF*H*W, lay_in.Cp] -> ([NF*Ho*Wo, 2 hp], Ho, Wo)"""
        h, hp, Cin = pk["h"], pk["hp"], pk["lay_in"].Cp
        P = NF * H * W
        if pk["down"]:
            d1, Ho, Wo = self.dw(x, NF, H, W, Cin, pk["wd1"], None, 2)
            Po = NF * Ho * Wo
            wf, bias = self.fold(pk["w1"], self.stats(d1, Po, pk["bn1a"]), pk["bn1a"])
            ya = self.pw(d1, Cin, Cin, wf, bias, Po)
            sa = (self.stats(ya, Po, pk["bn1b"]), pk["bn1b"])
            t = self.pw(x, Cin, Cin, pk["w2a"], None, P)
            stride, lda = 2, hp
        else:
            Ho, Wo, Po = H, W, P
            ya, sa, lda = x, None, 2 * hp                                   # first half of x: passed through
            t = self.pw(x[:, hp:], 2 * hp, hp, pk["w2a"], None, P)
            stride = 1
        d2, _, _ = self.dw(t, NF, H, W, hp, pk["wd2"], (self.stats(t, P, pk["bn2a"]), pk["bn2a"]), stride)
        wf, bias = self.fold(pk["w2c"], self.stats(d2, Po, pk["bn2b"]), pk["bn2b"])
        yb = self.pw(d2, hp, hp, wf, bias, Po)
        sb = (self.stats(yb, Po, pk["bn2c"]), pk["bn2c"])
        return self.shuffle(ya, lda, sa, yb, hp, sb, Po, h, hp), Ho, Wo


def trunk_forward(enc, x, NF, H, W):
    """x: the stem's output, channels-last rows [NF * H * W, 24] (bf16 or fp32) -> fp32 [NF, 1024]."""
    dt, dev = x.dtype, x.device
    packs, last = _packs(enc, dt, dev)
    if x.shape[1] != packs[0]["lay_in"].Cp:
        raise DttError(f"lipnet: the stem hands over {x.shape[1]} channels, the trunk expects {packs[0]['lay_in'].Cp}")
    run = _Run(dt, dev)
    for pk in packs:
        x, H, W = run.unit(pk, x, NF, H, W)
    P, lay = NF * H * W, last["lay_in"]
    y = run.pw(x, lay.Cp, lay.Cp, last["w"], None, P)
    b = last["bn"]
    st = run.stats(y, P, b)
    out = torch.empty((NF, y.shape[1]), dtype=F32, device=dev)
    check(lib().dtt_bn_relu_avgpool(ptr(y), ptr(st), ptr(b.gamma), ptr(b.beta), ptr(out), NF, H * W, y.shape[1], run.d,
                                    stream_ptr()), "dtt_bn_relu_avgpool")
    if run.tracked:
        torch._foreach_add_(run.tracked, 1)
    return out
