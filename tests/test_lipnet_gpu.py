"""The frozen ShuffleNetV2 lip-reading front end on the HIP kernels (lipnet.py / lipunit.hip) against the same module on
the stock ROCm modules (which tests/test_av_gpu.py::test_lipreading_front_end pins to the reference's fixture), the
unit kernels against their torch statements, and the time resampling of the video embedding against F.interpolate.
Reference: lipreading/models/shufflenetv2.py:26-104,151-163, lipreading/model.py:126-131,164-174,
DTTNetPlus/blocks/block_fusion.py:18-19."""
import copy
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from oracle.fill import synth_video
from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _encoder(seed):
    from dtt_amd.video import LipReadingEncoder
    torch.manual_seed(seed)
    enc = LipReadingEncoder().to(DEV)
    with torch.no_grad():                          # statistics and affine terms away from their initial 0 / 1
        for m in enc.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    for p in enc.parameters():
        p.requires_grad = False
    return enc


def _run(enc, x, own, bf16=False):
    from dtt_amd import video
    keep = video._LIP_OWN
    video._LIP_OWN = own
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            return enc(x).float()
    finally:
        video._LIP_OWN = keep


@pytest.mark.parametrize("train", [True, False])
def test_lip_front_end_on_own_kernels_equals_the_stock_modules(train):
    enc = _encoder(5)
    ref = copy.deepcopy(enc)
    enc.train(train)
    ref.train(train)
    x = synth_video(2, 6, seed=41).to(DEV).reshape(4, 1, 6, 88, 88)
    assert enc._own_kernels(x) is False                       # gradient mode on: stock path
    with torch.no_grad():
        assert enc._own_kernels(x)
    for step in range(2):                                     # twice: the second pass starts from the updated buffers
        y, yr = _run(enc, x, True), _run(ref, x, False)
        assert y.shape == yr.shape == (4, 6, 1024)
        e = rel_err(y, yr)
        assert e < 2e-4, (step, e)
    bufs, rbufs = dict(enc.named_buffers()), dict(ref.named_buffers())
    assert set(bufs) == set(rbufs)
    for k, b in bufs.items():
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(rbufs[k]) == (2 if train else 0), k
        else:
            e = rel_err(b, rbufs[k])
            assert e < 2e-4, (k, e)


def test_lip_front_end_bf16_autocast_stays_close_to_fp32():
    enc = _encoder(6).eval()
    x = synth_video(2, 5, seed=42).to(DEV).reshape(4, 1, 5, 88, 88)
    y32, y16 = _run(enc, x, True), _run(enc, x, True, bf16=True)
    stock16 = _run(enc, x, False, bf16=True)
    e_own, e_stock = rel_err(y16, y32), rel_err(stock16, y32)
    # frozen statistics; bf16 activations through 50-odd layers: bounded by twice the stock bf16 path's own error
    assert e_own < max(2 * e_stock, 2e-2), (e_own, e_stock)


def test_packs_follow_load_state_dict():
    enc = _encoder(7).eval()
    x = synth_video(1, 3, seed=43).to(DEV).reshape(2, 1, 3, 88, 88)
    y0 = _run(enc, x, True)
    other = _encoder(8).eval()
    enc.load_state_dict(other.state_dict())
    y1, want = _run(enc, x, True), _run(other, x, False)
    assert rel_err(y1, want) < 2e-4 and rel_err(y0, want) > 1e-2


def _lib():
    from dtt_amd import _lib
    return _lib


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-6), (torch.bfloat16, 6e-3)])
@pytest.mark.parametrize("stride,pre", [(1, True), (2, True), (2, False)])
def test_depthwise_kernel(dt, tol, stride, pre):
    L = _lib()
    torch.manual_seed(3)
    N, H, W, Cc = 3, 11, 7, 64
    x = torch.randn(N, H, W, Cc, device=DEV).to(dt)
    w = torch.randn(Cc, 1, 3, 3, device=DEV) * 0.3
    stats = torch.stack([torch.randn(Cc, device=DEV) * 0.2, torch.rand(Cc, device=DEV) + 0.5], 1).contiguous()
    gamma, beta = torch.rand(Cc, device=DEV) + 0.5, torch.randn(Cc, device=DEV) * 0.2
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty(N, Ho, Wo, Cc, device=DEV, dtype=dt)
    w9c = w.reshape(Cc, 9).t().contiguous()
    args = (L.ptr(stats), L.ptr(gamma), L.ptr(beta)) if pre else (None, None, None)
    L.check(L.lib().dtt_dwconv3x3_bnrelu(L.ptr(x), L.ptr(w9c), *args, L.ptr(y), N, H, W, Cc, stride, L.dtype_code(dt),
                                         L.stream_ptr()), "dw")
    xin = x.float()
    if pre:
        xin = torch.relu((xin - stats[:, 0]) * stats[:, 1] * gamma + beta)
    want = F.conv2d(xin.permute(0, 3, 1, 2), w, None, stride, 1, 1, Cc).permute(0, 2, 3, 1)
    assert rel_err(y.float(), want) < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("h,passthrough", [(58, True), (116, False), (232, True), (6, False)])
def test_bn_relu_shuffle_kernel(dt, h, passthrough):
    L = _lib()
    torch.manual_seed(4)
    P, hp = 777, (h + 7) // 8 * 8
    lda = 2 * hp if passthrough else hp
    a = torch.randn(P, lda, device=DEV).to(dt)
    b = torch.randn(P, hp, device=DEV).to(dt)

    def bn():
        return (torch.stack([torch.randn(hp, device=DEV) * 0.2, torch.rand(hp, device=DEV) + 0.5], 1).contiguous(),
                torch.rand(hp, device=DEV) + 0.5, torch.randn(hp, device=DEV) * 0.2)
    sa, sb = bn(), bn()
    out = torch.full((P, 2 * hp), 7.0, device=DEV).to(dt)
    pa = (None, None, None) if passthrough else tuple(L.ptr(t) for t in sa)
    L.check(L.lib().dtt_bn_relu_shuffle(L.ptr(a), lda, *pa, L.ptr(b), hp, *(L.ptr(t) for t in sb), L.ptr(out), C.c_int64(P),
                                        h, hp, L.dtype_code(dt), L.stream_ptr()), "shuffle")

    def act(v, s):
        return torch.relu((v.float()[:, :h] - s[0][:h, 0]) * s[0][:h, 1] * s[1][:h] + s[2][:h])
    va = a.float()[:, :h] if passthrough else act(a, sa)
    vb = act(b, sb)
    logical = torch.stack((va, vb), 2).reshape(P, 2 * h)              # channel_shuffle(cat(a, b), 2)
    want = torch.zeros(P, 2 * hp, device=DEV)
    want[:, :h], want[:, hp:hp + h] = logical[:, :h], logical[:, h:]
    tol = 1e-6 if dt == torch.float32 else 4e-3
    err, top = float((out.float() - want).abs().max()), max(1.0, float(want.abs().max()))
    assert err <= tol * top, (err, top)
    if passthrough:                                                                # the untouched half: bit-exact copies
        assert torch.equal(out[:, :hp].float()[:, 0:h:2], a.float()[:, :(h + 1) // 2])
    assert torch.equal(out.float()[:, h:hp], torch.zeros_like(want[:, h:hp]))      # padding columns are zeros


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-6), (torch.bfloat16, 4e-3)])
def test_fold_and_avgpool_kernels(dt, tol):
    L = _lib()
    torch.manual_seed(5)
    N, K = 120, 128
    w = torch.randn(N, K, device=DEV) * 0.1
    stats = torch.stack([torch.randn(K, device=DEV) * 0.2, torch.rand(K, device=DEV) + 0.5], 1).contiguous()
    gamma, beta = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.2
    wout, bias = torch.empty(N, K, device=DEV, dtype=dt), torch.empty(N, device=DEV)
    L.check(L.lib().dtt_pw_fold_bn(L.ptr(w), L.ptr(stats), L.ptr(gamma), L.ptr(beta), L.ptr(wout), L.ptr(bias), N, K,
                                   L.dtype_code(dt), L.stream_ptr()), "fold")
    sc = gamma * stats[:, 1]
    sh = beta - stats[:, 0] * sc
    assert rel_err(wout.float(), w * sc) < tol and rel_err(bias, w @ sh) < 1e-5
    NF, S, Cc = 37, 9, 1024
    y = torch.randn(NF, S, Cc, device=DEV).to(dt)
    st = torch.stack([torch.randn(Cc, device=DEV) * 0.2, torch.rand(Cc, device=DEV) + 0.5], 1).contiguous()
    g, b = torch.rand(Cc, device=DEV) + 0.5, torch.randn(Cc, device=DEV) * 0.2
    out = torch.empty(NF, Cc, device=DEV)
    L.check(L.lib().dtt_bn_relu_avgpool(L.ptr(y), L.ptr(st), L.ptr(g), L.ptr(b), L.ptr(out), NF, S, Cc, L.dtype_code(dt),
                                        L.stream_ptr()), "avgpool")
    want = torch.relu((y.float() - st[:, 0]) * st[:, 1] * g + b).mean(1)
    assert rel_err(out, want) < 1e-6


@pytest.mark.parametrize("Tin,Tout", [(50, 16), (50, 126), (50, 50), (7, 33), (1, 5), (64, 3)])
def test_lerp_time_against_interpolate(Tin, Tout):
    from dtt_amd import ops
    torch.manual_seed(Tin * 131 + Tout)
    x = torch.randn(3, Tin, 40, device=DEV, requires_grad=True)
    g = torch.randn(3, Tout, 40, device=DEV)
    y = ops.LerpTime.apply(x, Tout)
    y.backward(g)
    xr = x.detach().clone().requires_grad_(True)
    yr = F.interpolate(xr.transpose(-1, -2), size=Tout, mode="linear").transpose(-1, -2)
    yr.backward(g)
    assert (y - yr).abs().max() < 1e-6 and (x.grad - xr.grad).abs().max() < 1e-5
