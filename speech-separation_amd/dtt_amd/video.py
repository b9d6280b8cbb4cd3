"""Lip-ROI video encoders.  Same parameter names as the
reference and NO network fetch (the reference downloads a checkpoint with gdown in __init__,
resnet_video_encoder.py:46-60).  ResNet-18 encoder: every stage runs on the HIP kernels -- the front-end convolution
(dtt_stack_frames + ops.FrontConv), BatchNorm + Swish (+ residual) (ops.BatchNormAct), the max pool
(ops.MaxPool3x3s2), the trunk's 3x3 convolutions of either stride (ops.Conv3x3) and its 1x1 stride-2 downsample
convolutions (ops.Conv1x1s2): no library convolution is left in the step.  Switches for A/B runs:
DTT_VIDEO_OWN_CONV=0 (all convolutions on MIOpen), DTT_VIDEO_FRONT_CONV=0, DTT_VIDEO_S2_CONV=0 (only those).
ShuffleNetV2 encoder (FiLM model, frozen): the same front-end kernels, then the units on lipnet.py / lipunit.hip
(DTT_LIP_OWN=0: stock modules).

Reference: src/model/VideoEncoders/{resnet,resnet_video_encoder}.py.
"""
from __future__ import annotations

import math
import os

import torch
from torch import nn


_NATIVE_BN = os.environ.get("DTT_VIDEO_NATIVE_BN", "1") == "1"
_FUSED_BN = os.environ.get("DTT_VIDEO_FUSED_BN", "1") == "1"      # 0: ATen BatchNorm + elementwise Swish (A/B switch)
_OWN_CONV = os.environ.get("DTT_VIDEO_OWN_CONV", "1") == "1"      # 0: every lip-encoder convolution on MIOpen (A/B switch)
_OWN_FRONT = os.environ.get("DTT_VIDEO_FRONT_CONV", "1") == "1"   # 0: only the front-end convolution on MIOpen


_FUSED_POOL = os.environ.get("DTT_VIDEO_FUSED_POOL", "1") == "1"   # 0: front-end BatchNorm + Swish and max pool as two ops
_OWN_S2 = os.environ.get("DTT_VIDEO_S2_CONV", "1") == "1"         # 0: the trunk's stride-2 3x3 and 1x1 convolutions on MIOpen
_LIP_OWN = os.environ.get("DTT_LIP_OWN", "1") == "1"              # 0: the ShuffleNetV2 lip front end on the stock modules (A/B switch)


def _conv3x3(conv, x, slot=None):
    """The trunk's convolutions on the HIP kernels: 3x3 stride 1 / 2 (ops.Conv3x3) and the 1x1 stride-2 downsample
    (ops.Conv1x1s2); anything else on MIOpen.  slot: ops.GradSlot shared with x's other consumer in the block."""
    ok = (_OWN_CONV and conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0 and conv.bias is None
          and x.dtype in (torch.bfloat16, torch.float32))
    k, st, pd = conv.kernel_size, conv.stride, conv.padding
    if ok and ((k == (3, 3) and pd == (1, 1) and (st == (1, 1) or (st == (2, 2) and _OWN_S2)))
               or (k == (1, 1) and pd == (0, 0) and st == (2, 2) and _OWN_S2)):
        from .ops import Conv3x3, Conv1x1s2
        if torch.is_autocast_enabled("cuda") and x.dtype == torch.float32:
            x = x.to(torch.get_autocast_dtype("cuda"))
        if k == (1, 1):
            return Conv1x1s2.apply(x, conv.weight, slot)
        return Conv3x3.apply(x, conv.weight, st[0], slot)
    return conv(x)


def _slot_ok(conv, x):
    """True when _conv3x3 will run this convolution on the HIP kernels (only those take part in a GradSlot)."""
    k, st, pd = conv.kernel_size, conv.stride, conv.padding
    return (_OWN_CONV and conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0 and conv.bias is None
            and x.dtype in (torch.bfloat16, torch.float32)
            and ((k == (3, 3) and pd == (1, 1) and (st == (1, 1) or (st == (2, 2) and _OWN_S2)))
                 or (k == (1, 1) and pd == (0, 0) and st == (2, 2) and _OWN_S2)))


class _NativeBN:
    """BatchNorm through ATen's own channels-last kernels: MIOpen's NHWC batch-norm kernels took 25 ms of the
    107 ms cfg4 step (10x the HBM time of the tensors)."""

    def forward(self, x):
        if not _NATIVE_BN:
            return super().forward(x)
        with torch.backends.cudnn.flags(enabled=False):
            return super().forward(x)


class BatchNorm2d(_NativeBN, nn.BatchNorm2d):
    pass


class BatchNorm3d(_NativeBN, nn.BatchNorm3d):
    pass


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class BasicBlock(nn.Module):
    """resnet.py:36-81 with relu_type='swish'."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.relu1 = Swish()
        self.relu2 = Swish()
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        if not (_FUSED_BN and x.is_cuda):
            r = x if self.downsample is None else self.downsample(x)
            y = self.relu1(self.bn1(self.conv1(x)))
            y = self.bn2(self.conv2(y))
            return self.relu2(y + r)
        # BatchNorm + Swish (+ residual) as ONE HIP pass forward and two backward (ops.BatchNormAct); the stride-1
        # 3x3 convolutions on our kernels, the stride-2 / 1x1 ones on MIOpen
        # x has two consumers (conv1 and the residual / downsample branch): their partial gradients meet inside
        # conv1's data-gradient kernel instead of an autograd add (ops.GradSlot).  conv1 is issued FIRST so that the
        # engine, which runs ready nodes in reverse order of creation, reaches the downsample conv's backward before
        # conv1's (either order is handled).
        from .ops import batch_norm_act, new_slot
        own = _slot_ok(self.conv1, x) and (self.downsample is None or _slot_ok(self.downsample[0], x))
        slot = new_slot(x) if own else None
        y = _conv3x3(self.conv1, x, slot)
        if self.downsample is None:
            r, rslot = x, slot
        else:
            r, rslot = batch_norm_act(_conv3x3(self.downsample[0], x, slot), self.downsample[1], 0), None
        y = batch_norm_act(y, self.bn1, 1)
        return batch_norm_act(_conv3x3(self.conv2, y), self.bn2, 1, res=r, slot=rslot)


class ResNet18Trunk(nn.Module):
    """resnet.py:84-137, layers [2,2,2,2]."""

    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.layer1 = self._make(64, 2, 1)
        self.layer2 = self._make(128, 2, 2)
        self.layer3 = self._make(256, 2, 2)
        self.layer4 = self._make(512, 2, 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):  # (BatchNorm2d above is a subclass)
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make(self, planes, blocks, stride):
        ds = None
        if stride != 1 or self.inplanes != planes:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride=stride, bias=False), BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, ds)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.avgpool(x).flatten(1)


def frontend_fused(frontend3D, x, act):
    """frontend3D (Conv3d + BatchNorm3d + activation + MaxPool3d; act 1 swish: resnet_video_encoder.py:33-44, 2 relu:
    lipreading/model.py:126-131) on the HIP path.  With one input channel, Conv3d((5,7,7), stride (1,2,2), padding
    (2,3,3)) is a per-frame Conv2d over the 5 neighbouring frames stacked as channels (zero frames beyond the clip's
    ends), so its output is born as [frames, Co, 44, 44] in channels-last memory: no im2col of a 3-d volume, no
    layout change towards the 2-d trunk.  BatchNorm3d's statistics over (N, D, H, W) are BatchNorm2d's over
    (N * D, H, W); BN + activation and the (1,3,3) max pool run on the fused kernels."""
    from .ops import batch_norm_act, batch_norm_act_pool, MaxPool3x3s2

    def bn_pool(y):
        if _FUSED_POOL:
            return batch_norm_act_pool(y, bn, act)
        return MaxPool3x3s2.apply(batch_norm_act(y, bn, act))
    conv, bn = frontend3D[0], frontend3D[1]
    Bp, _, T, H, W = x.shape
    kt = conv.kernel_size[0]
    w2 = conv.weight[:, 0]                                                    # [64, 5, 7, 7]
    cpad = (-kt) % 8                                                          # channel count to a multiple of 8
    if kt <= 8 and kt % 2 == 1 and x.dtype == torch.float32 and not x.requires_grad:
        # one pass: fp32 clip -> stacked, channel-padded, channels-last frames in the conv's dtype (the torch
        # route below is five copies of a 200 MB tensor: pad, unfold, channel pad, layout, autocast)
        from . import ops
        odt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
        if odt not in (torch.bfloat16, torch.float32):
            odt = torch.float32
        own = (_OWN_CONV and _OWN_FRONT and conv.kernel_size[1:] == (7, 7) and conv.stride == (1, 2, 2) and conv.padding[1:] == (3, 3)
               and H % 2 == 0 and W % 2 == 0)
        padl, padr = (3, 5) if own else (0, 0)         # own conv: zero columns instead of per-pixel predicates
        buf = torch.empty((Bp * T, H, padl + W + padr, 8), dtype=odt, device=x.device)
        ops.check(ops.lib().dtt_stack_frames(ops.ptr(x.contiguous()), ops.ptr(buf), Bp, T, H, W, kt, padl, padr,
                                             1 if odt == torch.bfloat16 else 0, ops.stream_ptr()), "dtt_stack_frames")
        if own:
            y = ops.FrontConv.apply(buf, conv.weight, H, W).permute(0, 3, 1, 2)   # NCHW view of channels-last memory
            return bn_pool(y), T
        x5 = buf.permute(0, 3, 1, 2)                                          # NCHW view of channels-last memory
    else:
        xp = nn.functional.pad(x[:, 0], (0, 0, 0, 0, kt // 2, kt // 2))        # [B', T + 4, H, W]
        x5 = xp.unfold(1, kt, 1).permute(0, 1, 4, 2, 3).reshape(Bp * T, kt, H, W)   # frames t-2 .. t+2 as channels
        if cpad:
            x5 = nn.functional.pad(x5, (0, 0, 0, 0, 0, cpad))
        x5 = x5.contiguous(memory_format=torch.channels_last)
    if cpad:
        w2 = nn.functional.pad(w2, (0, 0, 0, 0, 0, cpad))
    y = nn.functional.conv2d(x5, w2, None, stride=conv.stride[1:], padding=conv.padding[1:])
    return bn_pool(y), T


class VideoEncoder(nn.Module):
    """resnet_video_encoder.py:22-72: Conv3d(5,7,7)+BN3d+Swish+MaxPool3d -> ResNet-18.
    video [B, N, T, H, W] -> [B, N, 512, T]."""

    def __init__(self, checkpoint_path=None):
        super().__init__()
        self.frontend_nout, self.backend_out = 64, 512
        self.trunk = ResNet18Trunk()
        self.frontend3D = nn.Sequential(
            nn.Conv3d(1, 64, kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            BatchNorm3d(64), Swish(),
            nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)))
        if checkpoint_path is not None and os.path.exists(checkpoint_path):
            ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
            self.load_state_dict(ck["model_state_dict"], strict=False)

    def _frontend_fused(self, x):
        return frontend_fused(self.frontend3D, x, 1)

    def forward(self, x):
        B, N, T, H, W = x.shape
        if _FUSED_BN and x.is_cuda:
            x, Tn = self._frontend_fused(x.reshape(B * N, 1, T, H, W))
        else:
            x = self.frontend3D(x.reshape(B * N, 1, T, H, W))
            Tn = x.shape[2]
            x = x.transpose(1, 2).reshape(B * N * Tn, x.shape[1], x.shape[3], x.shape[4])
            x = x.contiguous(memory_format=torch.channels_last)
        x = self.trunk(x)
        return x.view(B, N, Tn, -1).transpose(2, 3)


# --------------------------------------------------------------------------------------------- #
# Lip-reading front end of the FiLM model (DTTNetPlus): Conv3d stem + ShuffleNetV2 (1.0x) trunk.
# Reference: src/model/DTTNetPlus/modules/video_encoder.py:8-47 (wrapper, frozen encoder, trainable
# post_processor), src/model/lipreading/model.py:106-185 (Lipreading with backbone_type="shufflenet",
# relu_type="relu", extract_feats=True: only frontend3D + trunk run), src/model/lipreading/models/
# shufflenetv2.py (units and stage widths).  Stock PyTorch-ROCm modules, channels-last, under the same
# parameter names (``encoder.frontend3D.*``, ``encoder.trunk.0.<unit>.banch1|banch2.*``,
# ``encoder.trunk.1.*``).  The TCN classification head of the reference's Lipreading object
# (``encoder.tcn.*``) is never executed with extract_feats=True and is not materialised here:
# checkpoints are exchanged with strict=False on both sides (model_loader.py:227-239).
# --------------------------------------------------------------------------------------------- #
def _pw(ci, co):
    return [nn.Conv2d(ci, co, 1, 1, 0, bias=False), BatchNorm2d(co), nn.ReLU(inplace=True)]


def _dw(c, stride):
    return [nn.Conv2d(c, c, 3, stride, 1, groups=c, bias=False), BatchNorm2d(c)]


class ShuffleUnit(nn.Module):
    """One ShuffleNetV2 unit (shufflenetv2.py:42-104).  ``banch1`` / ``banch2`` are the reference's
    (misspelt) attribute names: they are state_dict keys."""

    def __init__(self, inp, oup, stride):
        super().__init__()
        half = oup // 2
        self.down = stride == 2
        if self.down:
            self.banch1 = nn.Sequential(*_dw(inp, 2), *_pw(inp, half))
            self.banch2 = nn.Sequential(*_pw(inp, half), *_dw(half, 2), *_pw(half, half))
        else:
            self.banch2 = nn.Sequential(*_pw(half, half), *_dw(half, 1), *_pw(half, half))

    def forward(self, x):
        if self.down:
            a, b = self.banch1(x), self.banch2(x)
        else:
            a, b = x.chunk(2, dim=1)
            b = self.banch2(b)
        # concat + channel shuffle with 2 groups = interleave the two halves channel by channel
        n, c, h, w = a.shape
        return torch.stack((a, b), dim=2).reshape(n, 2 * c, h, w).contiguous(memory_format=torch.channels_last)


class LipReadingEncoder(nn.Module):
    """Lipreading(modality="video", backbone_type="shufflenet", width_mult=1.0, relu_type="relu",
    extract_feats=True): x [B', 1, T, 88, 88] -> [B', T, 1024]."""

    def __init__(self):
        super().__init__()
        widths = (24, 116, 232, 464)
        units, cin = [], widths[0]
        for cout, rep in zip(widths[1:], (4, 8, 4)):
            for i in range(rep):
                units.append(ShuffleUnit(cin, cout, 2 if i == 0 else 1))
                cin = cout
        self.trunk = nn.Sequential(nn.Sequential(*units), nn.Sequential(*_pw(cin, 1024)),
                                   nn.Sequential(nn.AvgPool2d(3)))
        self.frontend3D = nn.Sequential(
            nn.Conv3d(1, widths[0], kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            BatchNorm3d(widths[0]), nn.ReLU(True),
            nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)))
        for m in self.modules():                               # model.py:187-211
            if isinstance(m, (nn.Conv3d, nn.Conv2d)):
                n = m.out_channels
                for k in m.kernel_size:
                    n *= k
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, (nn.BatchNorm3d, nn.BatchNorm2d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _own_kernels(self, x):
        """The HIP path takes the reference's geometry (one-channel fp32 clips whose trunk output is 3 x 3: 88 x 88 lip
        crops) when no gradient is wanted (the encoder is frozen, video_encoder.py:38-39); anything else runs on the
        stock modules below."""
        if not (_LIP_OWN and _FUSED_BN and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()):
            return False
        H, W = x.shape[3], x.shape[4]
        if x.shape[1] != 1 or H % 2 or W % 2:
            return False
        for _ in range(5):                                    # stem conv, max pool, three stride-2 units
            H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        return (H, W) == (3, 3)

    def forward(self, x, lengths=None):
        Bp = x.shape[0]
        if self._own_kernels(x):
            from . import lipnet
            y, Tn = frontend_fused(self.frontend3D, x, 2)                       # [B' T, 24, 22, 22], channels-last memory
            NF, Cc, H, W = y.shape
            rows = y.permute(0, 2, 3, 1).contiguous().view(NF * H * W, Cc)
            return lipnet.trunk_forward(self, rows, NF, H, W).view(Bp, Tn, -1)
        x = self.frontend3D(x)
        Tn = x.shape[2]
        x = x.transpose(1, 2).reshape(Bp * Tn, x.shape[1], x.shape[3], x.shape[4])
        x = self.trunk(x.contiguous(memory_format=torch.channels_last))
        return x.reshape(Bp, Tn, 1024)


class VideoEncoderModule(nn.Module):
    """DTTNetPlus/modules/video_encoder.py:8-47: frozen lip-reading encoder + trainable Linear(2048 -> D).
    video [B, 2, T, 88, 88] -> [B, T, D].  As in the reference the encoder's parameters are frozen but its
    BatchNorm layers follow the module's train()/eval() mode."""

    def __init__(self, load_weights_path=None, video_embed_dim=256):
        super().__init__()
        self.encoder = LipReadingEncoder()
        self.load_weights_path = load_weights_path
        self.post_processor = nn.Linear(1024 * 2, video_embed_dim)
        if load_weights_path is not None:
            if not os.path.isfile(load_weights_path):
                raise FileNotFoundError(f"video encoder weights not found: {load_weights_path}")
            ck = torch.load(load_weights_path, map_location="cpu", weights_only=True)
            missing = self.encoder.load_state_dict(ck["model_state_dict"], strict=False)
            lost = [k for k in missing.missing_keys]
            if lost:
                raise KeyError(f"video encoder checkpoint lacks {lost[:3]}...")
        for p in self.encoder.parameters():
            p.requires_grad = False

    def features(self, video, bf16=False):
        """[B, 2, T, H, W] -> [B, T, 2048] lip features (no gradient: frozen)."""
        B, Cn, T, H, W = video.shape
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            out = self.encoder(video.reshape(B * Cn, 1, T, H, W)).reshape(B, Cn, T, 1024)
        return out.transpose(1, 2).flatten(-2, -1).float()

    def forward(self, video, bf16=False):
        feats = self.features(video, bf16)
        if feats.is_cuda:                      # the trainable projection (video_encoder.py:33,46) through dtt_gemm, fp32
            from .ops import Linear
            with torch.autocast("cuda", enabled=False):
                return Linear.apply(feats, self.post_processor.weight, self.post_processor.bias, False)
        return self.post_processor(feats)
