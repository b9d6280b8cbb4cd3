"""dtt_gemm on the LDS-DMA tile kernels of csrc/gemmx.hip (gathered 3x3 stride-1 conv rows and plain k-contiguous
products, every workgroup tile) against fp64 CPU references of the same op (torch conv2d / matmul on the same
bf16-rounded operands): the Conv2d(c, c, 3, padding="same") of block_tfc_tdf.py:12-17 and the ResNet-18 trunk's
conv3x3 (VideoEncoders/resnet.py:26-29) at image sizes from 3 x 3 to 257 x 63, ragged last tiles, bias and residual."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _launches():
    from dtt_amd._lib import lib
    fn = lib().dtt_gemmx_launches
    fn.restype = __import__("ctypes").c_longlong
    return fn()


@pytest.mark.parametrize("B,Fq,T,Ci,Co,res", [
    (2, 65, 16, 512, 512, False),        # latent TFC conv: 128 x 128 tiles, K = 4608
    (3, 129, 32, 128, 128, True),        # level-2 conv with a fused residual (GradSlot data gradient)
    (1, 257, 63, 64, 64, False),         # 256 x 64 tiles
    (70, 6, 6, 256, 256, False),         # lip encoder layer3: 6 x 6 images, every tile spans 7 frames
    (300, 3, 3, 512, 512, True),         # layer4: 3 x 3 images (the centre pixel is the only one with nine taps)
    (20, 11, 11, 128, 128, False),       # layer2
    (16, 22, 22, 64, 128, False),        # Ci != Co
    (9, 31, 29, 96, 160, False),         # channel counts that are multiples of 32 only; ragged M and N tiles
])
def test_conv_rows_against_fp64(B, Fq, T, Ci, Co, res):
    from dtt_amd import ops
    g = torch.Generator().manual_seed(Ci + Fq)
    x = torch.randn(B, Fq, T, Ci, generator=g).to(torch.bfloat16)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3.0 * Ci ** 0.5)).to(torch.bfloat16)
    bias = torch.randn(Co, generator=g)
    r = torch.randn(B, Fq, T, Co, generator=g).to(torch.bfloat16) if res else None
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    if res:
        ref = ref + r.double()
    xd = x.to(DEV)
    wp = w.to(DEV).permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()          # [Co][tap][Ci]
    out = torch.full((B, Fq, T, Co), float("nan"), dtype=torch.bfloat16, device=DEV)
    geom = dict(cB=B, cFi=Fq, cTi=T, cCi=Ci, cFo=Fq, cTo=T, cstride=1, ctrans=0)
    n0 = _launches()
    ops.conv_gemm(xd, wp, bias.to(DEV), out, geom, res=None if r is None else r.to(DEV))
    torch.cuda.synchronize()
    assert _launches() == n0 + 1, "the shape did not run on gemmx"
    o = out.double().cpu()
    assert torch.isfinite(o).all()
    err = float((o - ref).norm() / ref.norm())
    worst = float((o - ref).abs().max() / ref.abs().max())
    assert err < 4e-3 and worst < 2e-2, (err, worst)          # bf16 output rounding: 2^-9 relative per element


@pytest.mark.parametrize("M,N,K,res,f32,here", [
    (33280, 256, 1024, True, False, False),    # TC_FC_RNN output Linear (latent.py:93-96) + residual: gemm256's shape
    (16640, 512, 256, True, False, False),     # 130 tiles of 256 x 256: still gemm256 (its threshold is 128 tiles)
    (5000, 512, 96, False, False, True),       # K = 3 k-steps: shortest pipeline
    (4100, 136, 256, False, True, True),       # ragged M and N, fp32 output
    (2048, 64, 512, False, False, False),      # N = 64 plain products stay on gemm_fast_kernel
])
def test_plain_products_against_fp64(M, N, K, res, f32, here):
    from dtt_amd._lib import gemm_raw
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g).to(torch.bfloat16) if res else None
    ref = a.double() @ b.double().t() + bias.double()
    if res:
        ref = ref + r.double()
    out = torch.full((M, N), float("nan"), dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
    kw = dict(A=a.to(DEV), B=b.to(DEV), C=out, bias=bias.to(DEV), bias_mode=1, M=M, N=N, K=K, a_ms=K, a_ks=1, b_ks=1,
              b_ns=K, c_ms=N, c_ns=1, dtype=1, out_f32=1 if f32 else 0)
    if res:
        kw.update(R=r.to(DEV), r_ms=N, r_ns=1)
    n0 = _launches()
    gemm_raw(**kw)
    torch.cuda.synchronize()
    ran_here = _launches() == n0 + 1
    o = out.double().cpu()
    assert torch.isfinite(o).all()
    err = float((o - ref).norm() / ref.norm())
    assert err < (1e-5 if f32 else 4e-3), err
    assert ran_here == here, "kernel selection changed: update the table above"
