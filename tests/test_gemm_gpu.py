"""GPU: dtt_gemm (MFMA GEMM + gather loaders) against fp64 CPU math.
fp32 mode uses the exact-f32 MFMA: tolerance 1e-5 relative; bf16: 1e-2."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from helpers import rel_err  # noqa: E402

TOL = {torch.float32: 1e-5, torch.bfloat16: 1e-2}


def _mk(shape, dt, g):
    return torch.randn(shape, generator=g).to(dt)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (513, 70, 257), (33, 300, 520), (200, 31, 40),
                                   (64, 64, 1032), (1, 1, 8)])
def test_gemm_nt_nn_tn(dt, M, N, K):
    from dtt_amd._lib import gemm_raw, dtype_code
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    dev = "cuda"
    a = _mk((M, K), dt, g)
    b = _mk((K, N), dt, g)
    bias_n = torch.randn(N, generator=g)
    bias_m = torch.randn(M, generator=g)
    r = _mk((M, N), dt, g)
    ref = a.double() @ b.double()
    ad, bd, rd = a.to(dev), b.to(dev), r.to(dev)
    # NT: A[m,k] k-contig, B stored [n,k] k-contig
    bt = b.t().contiguous().to(dev)
    c = torch.empty(M, N, dtype=dt, device=dev)
    gemm_raw(A=ad, B=bt, C=c, bias=bias_n.to(dev), R=rd, M=M, N=N, K=K, a_ms=K, a_ks=1, b_ks=1, b_ns=K,
             c_ms=N, c_ns=1, r_ms=N, r_ns=1, dtype=dtype_code(dt), bias_mode=1)
    assert rel_err(c, ref + bias_n.double() + r.double()) < TOL[dt]
    # NN: B[k,n] n-contig, per-m bias, fp32 output
    c = torch.empty(M, N, dtype=torch.float32, device=dev)
    gemm_raw(A=ad, B=bd, C=c, bias=bias_m.to(dev), M=M, N=N, K=K, a_ms=K, a_ks=1, b_ks=N, b_ns=1,
             c_ms=N, c_ns=1, dtype=dtype_code(dt), bias_mode=2, out_f32=1)
    assert rel_err(c, ref + bias_m.double()[:, None]) < TOL[dt]
    # TN: A stored [k,m] m-contig, transposed output (c_ms = 1), GELU
    at = a.t().contiguous().to(dev)
    c = torch.empty(N, M, dtype=dt, device=dev)
    gemm_raw(A=at, B=bd, C=c, M=M, N=N, K=K, a_ms=1, a_ks=M, b_ks=N, b_ns=1, c_ms=1, c_ns=M,
             dtype=dtype_code(dt), act=1, alpha=0.5)
    assert rel_err(c.t(), F.gelu(0.5 * ref)) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(513, 300, 257), (64, 200, 513), (20, 72, 100), (300, 20, 64), (130, 64, 61)])
def test_gemm_fast_path_padded_operands(dt, M, N, K):
    """The compile-time-layout kernels: rows padded to 16 bytes with NaN in the padding (ragged K
    chunks must be masked, ragged row chunks must never reach C), residual, bias, GELU, fp32 out,
    batched, and the atomic split-K form."""
    from dtt_amd._lib import gemm_raw, dtype_code
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    dev = "cuda"
    pad = lambda v: (v + 15) // 16 * 16
    Kp, Np, Mp = pad(K), pad(N), pad(M)
    Bn = 2
    a = _mk((Bn, M, K), dt, g)
    b = _mk((Bn, K, N), dt, g)
    r = _mk((Bn, M, N), dt, g)
    bias_n, bias_m = torch.randn(N, generator=g), torch.randn(M, generator=g)
    ref = a.double() @ b.double()
    nanfill = lambda shape: torch.full(shape, float("nan"), dtype=dt, device=dev)
    a_kc = nanfill((Bn, M, Kp)); a_kc[:, :, :K] = a.to(dev)               # A[m][k], k contiguous
    a_rc = nanfill((Bn, K, Mp)); a_rc[:, :, :M] = a.transpose(1, 2).to(dev)  # A[k][m], m contiguous
    b_kc = nanfill((Bn, N, Kp)); b_kc[:, :, :K] = b.transpose(1, 2).to(dev)  # B[n][k]
    b_rc = nanfill((Bn, K, Np)); b_rc[:, :, :N] = b.to(dev)               # B[k][n]
    rp = nanfill((Bn, M, Np)); rp[:, :, :N] = r.to(dev)
    d = dtype_code(dt)
    # KC x KC, bias per n, residual
    c = nanfill((Bn, M, Np))
    gemm_raw(A=a_kc, B=b_kc, C=c, bias=bias_n.to(dev), R=rp, M=M, N=N, K=K, batch=Bn, a_ms=Kp, a_ks=1, a_bs=M * Kp,
             b_ks=1, b_ns=Kp, b_bs=N * Kp, c_ms=Np, c_ns=1, c_bs=M * Np, r_ms=Np, r_ns=1, r_bs=M * Np, dtype=d,
             bias_mode=1)
    assert rel_err(c[:, :, :N], ref + bias_n.double() + r.double()) < TOL[dt]
    assert torch.isnan(c[:, :, N:]).all()                    # padding of C untouched
    # KC x RC, bias per m, GELU, fp32 output
    c = torch.full((Bn, M, Np), float("nan"), device=dev)
    gemm_raw(A=a_kc, B=b_rc, C=c, bias=bias_m.to(dev), M=M, N=N, K=K, batch=Bn, a_ms=Kp, a_ks=1, a_bs=M * Kp,
             b_ks=Np, b_ns=1, b_bs=K * Np, c_ms=Np, c_ns=1, c_bs=M * Np, dtype=d, bias_mode=2, act=1, out_f32=1)
    assert rel_err(c[:, :, :N], F.gelu(ref + bias_m.double()[:, None])) < TOL[dt]
    # RC x RC plain store
    c = nanfill((Bn, M, Np))
    gemm_raw(A=a_rc, B=b_rc, C=c, M=M, N=N, K=K, batch=Bn, a_ms=1, a_ks=Mp, a_bs=K * Mp, b_ks=Np, b_ns=1,
             b_bs=K * Np, c_ms=Np, c_ns=1, c_bs=M * Np, dtype=d)
    assert rel_err(c[:, :, :N], ref) < TOL[dt]
    # RC x RC, batches and K-splits accumulated atomically into one fp32 C
    c = torch.zeros(M, N, device=dev)
    gemm_raw(A=a_rc, B=b_rc, C=c, M=M, N=N, K=K, batch=Bn, splitk=3, a_ms=1, a_ks=Mp, a_bs=K * Mp, b_ks=Np,
             b_ns=1, b_bs=K * Np, c_ms=N, c_ns=1, c_bs=0, dtype=d, atomic=1, out_f32=1)
    assert rel_err(c, ref.sum(0)) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_batched_splitk_atomic(dt):
    from dtt_amd._lib import gemm_raw, dtype_code
    g = torch.Generator().manual_seed(5)
    dev = "cuda"
    Bn, M, N, K = 3, 24, 40, 1000
    a = _mk((Bn, K, M), dt, g)           # [k, m] per batch: TN wgrad form
    b = _mk((Bn, K, N), dt, g)
    ref = torch.einsum("bkm,bkn->mn", a.double(), b.double())       # reduce over batch too
    c = torch.zeros(M, N, dtype=torch.float32, device=dev)
    gemm_raw(A=a.to(dev), B=b.to(dev), C=c, M=M, N=N, K=K, batch=Bn, splitk=4, a_ms=1, a_ks=M, a_bs=K * M,
             b_ks=N, b_ns=1, b_bs=K * N, c_ms=N, c_ns=1, c_bs=0, dtype=dtype_code(dt), atomic=1, out_f32=1)
    assert rel_err(c, ref) < TOL[dt]
    # batched, distinct outputs
    c = torch.empty(Bn, M, N, dtype=dt, device=dev)
    gemm_raw(A=a.to(dev), B=b.to(dev), C=c, M=M, N=N, K=K, batch=Bn, a_ms=1, a_ks=M, a_bs=K * M,
             b_ks=N, b_ns=1, b_bs=K * N, c_ms=N, c_ns=1, c_bs=M * N, dtype=dtype_code(dt))
    assert rel_err(c, torch.einsum("bkm,bkn->bmn", a.double(), b.double())) < TOL[dt]


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Ci,Co,Fi,Ti,stride", [(8, 16, 9, 7, 1), (32, 32, 33, 20, 1), (16, 32, 17, 13, 2),
                                                (64, 128, 10, 6, 2)])
def test_conv_implicit_gemm_fwd_and_wgrad(dt, Ci, Co, Fi, Ti, stride):
    from dtt_amd._lib import gemm_raw, dtype_code
    g = torch.Generator().manual_seed(Ci + Co + Fi)
    dev = "cuda"
    Bn = 2
    x = _mk((Bn, Ci, Fi, Ti), dt, g)
    w = (_mk((Co, Ci, 3, 3), dt, g).float() / (3 * Ci ** 0.5)).to(dt)
    bias = torch.randn(Co, generator=g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride=stride, padding=1)
    Fo, To = ref.shape[2], ref.shape[3]
    wp = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()     # [co][tap][ci]
    xd = _nhwc(x).to(dev)
    y = torch.empty(Bn, Fo, To, Co, dtype=dt, device=dev)
    geom = dict(cB=Bn, cFi=Fi, cTi=Ti, cCi=Ci, cFo=Fo, cTo=To, cstride=stride, ctrans=0)
    gemm_raw(A=xd, B=wp.to(dev), C=y, bias=bias.to(dev), M=Bn * Fo * To, N=Co, K=9 * Ci, a_ms=0, a_ks=1,
             b_ks=1, b_ns=9 * Ci, c_ms=Co, c_ns=1, dtype=dtype_code(dt), bias_mode=1, a_mode=1, **geom)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < TOL[dt]
    # wgrad: dW[co][tap][ci] = sum_p dy[p,co] * x[src(p,tap),ci]
    dy = _mk((Bn, Co, Fo, To), dt, g)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    F.conv2d(xr, wr, None, stride=stride, padding=1).backward(dy.double())
    dwp = torch.zeros(Co, 9, Ci, dtype=torch.float32, device=dev)
    P = Bn * Fo * To
    gemm_raw(A=_nhwc(dy).to(dev), B=xd, C=dwp, M=Co, N=Ci, K=P, batch=9, splitk=3, a_ms=1, a_ks=Co, a_bs=0,
             b_ks=Ci, b_ns=1, b_bs=0, c_ms=9 * Ci, c_ns=1, c_bs=Ci, dtype=dtype_code(dt), atomic=1,
             out_f32=1, b_mode=1, **geom)
    ref_dw = wr.grad.permute(0, 2, 3, 1).reshape(Co, 9, Ci)
    assert rel_err(dwp, ref_dw) < TOL[dt]
    # dgrad through the same kernel: stride 1 -> flipped-weight conv; stride 2 -> transposed gather
    if stride == 1:
        wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Ci, 9 * Co).contiguous()     # [ci][tap'][co]
        gd = dict(cB=Bn, cFi=Fo, cTi=To, cCi=Co, cFo=Fi, cTo=Ti, cstride=1, ctrans=0)
    else:
        wd = w.permute(1, 2, 3, 0).reshape(Ci, 9 * Co).contiguous()
        gd = dict(cB=Bn, cFi=Fo, cTi=To, cCi=Co, cFo=Fi, cTo=Ti, cstride=2, ctrans=1)
    dx = torch.empty(Bn, Fi, Ti, Ci, dtype=dt, device=dev)
    gemm_raw(A=_nhwc(dy).to(dev), B=wd.to(dev), C=dx, M=Bn * Fi * Ti, N=Ci, K=9 * Co, a_ms=0, a_ks=1, b_ks=1,
             b_ns=9 * Co, c_ms=Ci, c_ns=1, dtype=dtype_code(dt), a_mode=1, **gd)
    assert rel_err(dx.permute(0, 3, 1, 2), xr.grad) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_conv_transposed_cropped(dt):
    """ConvTranspose2d(3, stride 2, pad 1, output_padding 1) + crop (decoder.py:46-52)."""
    from dtt_amd._lib import gemm_raw, dtype_code
    g = torch.Generator().manual_seed(9)
    dev = "cuda"
    Bn, Ci, Co, Fi, Ti = 2, 32, 16, 9, 7
    Fs, Ts = 17, 13                                            # skip size (crop of 18 x 14)
    x = _mk((Bn, Ci, Fi, Ti), dt, g)
    w = (_mk((Ci, Co, 3, 3), dt, g).float() / (3 * Ci ** 0.5)).to(dt)
    ref = F.conv_transpose2d(x.double(), w.double(), None, stride=2, padding=1, output_padding=1)[:, :, :Fs, :Ts]
    wp = w.permute(1, 2, 3, 0).reshape(Co, 9 * Ci).contiguous()     # [co][tap][ci]
    y = torch.empty(Bn, Fs, Ts, Co, dtype=dt, device=dev)
    gemm_raw(A=_nhwc(x).to(dev), B=wp.to(dev), C=y, M=Bn * Fs * Ts, N=Co, K=9 * Ci, a_ms=0, a_ks=1, b_ks=1,
             b_ns=9 * Ci, c_ms=Co, c_ns=1, dtype=dtype_code(dt), a_mode=1, cB=Bn, cFi=Fi, cTi=Ti, cCi=Ci,
             cFo=Fs, cTo=Ts, cstride=2, ctrans=1)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Fi,Ti,Fo,To", [(9, 7, 18, 14), (9, 7, 17, 13), (5, 4, 9, 8)])
def test_conv_transposed_by_output_parity(dt, Fi, Ti, Fo, To):
    """ops.conv_gemm on a transposed geometry runs four parity-phase GEMMs (DttGemm::cphase): against
    F.conv_transpose2d (decoder up-sampling, even planes) and against the data gradient of the
    stride-2 conv (encoder down-sampling, odd planes: Fo = 2*Fi - 1)."""
    from dtt_amd import ops
    g = torch.Generator().manual_seed(Fi * 100 + Fo)
    dev = "cuda"
    Bn, Ci, Co = 2, 16, 24
    x = _mk((Bn, Ci, Fi, Ti), dt, g)
    w = _mk((Ci, Co, 3, 3), dt, g) * 0.2                     # ConvTranspose2d layout [Ci, Co, 3, 3]
    bias = torch.randn(Co, generator=g)
    full = F.conv_transpose2d(x.double(), w.double(), bias.double(), stride=2, padding=1, output_padding=1)
    ref = full[:, :, :Fo, :To]                                # odd planes: the last row/column is never produced
    wp = w.permute(1, 2, 3, 0).reshape(Co, 9 * Ci).contiguous().to(dev)
    out = torch.full((Bn, Fo, To, Co), float("nan"), dtype=dt, device=dev)
    geom = dict(cB=Bn, cFi=Fi, cTi=Ti, cCi=Ci, cFo=Fo, cTo=To, cstride=2, ctrans=1)
    ops.conv_gemm(x.permute(0, 2, 3, 1).contiguous().to(dev), wp, bias.to(dev), out, geom)
    assert not torch.isnan(out.float()).any()
    assert rel_err(out.permute(0, 3, 1, 2), ref) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_row_shift(dt):
    """dW_hh form: C[m][n] = sum_r dG[r][m] * y[r + shift][n], zero at sequence edges."""
    from dtt_amd._lib import gemm_raw, dtype_code
    g = torch.Generator().manual_seed(3)
    dev = "cuda"
    outer, L, heads, G, H = 3, 5, 2, 48, 24
    R = outer * L * heads
    dG = _mk((R, G), dt, g)
    y = _mk((R, 2 * H), dt, g)
    for shift_steps, edge, col0 in ((-1, 0, 0), (1, L - 1, H)):
        yy = y.double().view(outer, L, heads, 2 * H)[..., col0:col0 + H]
        prev = torch.zeros_like(yy)
        if shift_steps < 0:
            prev[:, 1:] = yy[:, :-1]
        else:
            prev[:, :-1] = yy[:, 1:]
        ref = dG.double().t() @ prev.reshape(R, H)
        c = torch.zeros(G, H, dtype=torch.float32, device=dev)
        yd = y.to(dev)
        gemm_raw(A=dG.to(dev), B=yd[:, col0:], C=c, M=G, N=H, K=R, splitk=2, a_ms=1, a_ks=G, b_ks=2 * H, b_ns=1,
                 c_ms=H, c_ns=1, dtype=dtype_code(dt), atomic=1, out_f32=1, b_mode=2,
                 sh_shift=shift_steps * heads, sh_heads=heads, sh_L=L, sh_edge=edge)
        assert rel_err(c, ref) < TOL[dt]


@pytest.mark.parametrize("Cc", [32, 64])
def test_conv3x3_persistent_kernels_walk_tiles_across_batch_items(Cc):
    """conv3x3_c32_kernel / conv3x3_c64_kernel (block_tfc_tdf.py:12-17 at the two upper U-Net levels): one workgroup
    per CU walks several tiles, and with 81 tiles per item and 2 per workgroup a walk straddles two batch items -- the
    running InstanceNorm sums must be flushed per item.  Output, residual form and statistics against fp64."""
    from dtt_amd import ops
    g = torch.Generator().manual_seed(700 + Cc)
    dev, dt = "cuda", torch.bfloat16
    Bn, Fq, T = 5, 130, 131
    x = _mk((Bn, Cc, Fq, T), dt, g)
    w = (_mk((Cc, Cc, 3, 3), dt, g).float() / (3 * Cc ** 0.5)).to(dt)
    bias = torch.randn(Cc, generator=g)
    res = _mk((Bn, Cc, Fq, T), dt, g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    wp = w.permute(0, 2, 3, 1).reshape(Cc, 9 * Cc).contiguous().to(dev)
    xd = _nhwc(x).to(dev)
    y, stats = ops.conv3x3_s1(xd, wp, bias.to(dev), want_stats=True)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < TOL[dt]
    assert rel_err(stats[..., 0], ref.mean(dim=(2, 3))) < 2e-3
    assert rel_err(stats[..., 1], 1.0 / torch.sqrt(ref.var(dim=(2, 3), unbiased=False) + 1e-5)) < 2e-3
    y2, _ = ops.conv3x3_s1(xd, wp, bias.to(dev), res=_nhwc(res).to(dev))
    assert rel_err(y2.permute(0, 3, 1, 2), ref + res.double()) < TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Ci,Co,Fq,T", [(32, 32, 33, 20), (64, 64, 17, 13), (32, 96, 16, 16), (128, 64, 9, 35),
                                          (64, 64, 22, 22), (64, 128, 38, 17), (128, 128, 65, 16)])
def test_conv3x3_dedicated_kernels(dt, Ci, Co, Fq, T):
    """dtt_conv3x3_s1 (+ fused residual and InstanceNorm statistics) and dtt_conv3x3_wgrad.  The row counts cover
    every length of the last tile row's walk: 1 valid row (17, 33, 65), 6 (22, 38), 9 and 16."""
    from dtt_amd import ops
    g = torch.Generator().manual_seed(Ci * 3 + Co + Fq)
    dev = "cuda"
    Bn = 3
    x = _mk((Bn, Ci, Fq, T), dt, g)
    w = (_mk((Co, Ci, 3, 3), dt, g).float() / (3 * Ci ** 0.5)).to(dt)
    bias = torch.randn(Co, generator=g)
    res = _mk((Bn, Co, Fq, T), dt, g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    wp = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous().to(dev)
    xd = _nhwc(x).to(dev)
    y, stats = ops.conv3x3_s1(xd, wp, bias.to(dev), want_stats=True)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < TOL[dt]
    mean = ref.mean(dim=(2, 3))
    rstd = 1.0 / torch.sqrt(ref.var(dim=(2, 3), unbiased=False) + 1e-5)
    assert rel_err(stats[..., 0], mean) < (1e-5 if dt == torch.float32 else 2e-3)
    assert rel_err(stats[..., 1], rstd) < (1e-5 if dt == torch.float32 else 2e-3)
    y2, _ = ops.conv3x3_s1(xd, wp, bias.to(dev), res=_nhwc(res).to(dev))
    assert rel_err(y2.permute(0, 3, 1, 2), ref + res.double()) < TOL[dt]
    if dt == torch.bfloat16:
        import ctypes as C
        from dtt_amd._lib import lib, check, ptr, stream_ptr
        dy = _mk((Bn, Co, Fq, T), dt, g)
        xr = x.double()
        wr = w.double().requires_grad_(True)
        F.conv2d(xr, wr, None, padding=1).backward(dy.double())
        dwp = torch.zeros(Co, 9, Ci, dtype=torch.float32, device=dev)
        check(lib().dtt_conv3x3_wgrad(ptr(_nhwc(dy).to(dev)), ptr(xd), ptr(dwp), Bn, Fq, T, Ci, Co, 1, stream_ptr()), "wgrad")
        assert rel_err(dwp, wr.grad.permute(0, 2, 3, 1).reshape(Co, 9, Ci)) < TOL[dt]
        # the workspace form (partial sums stored per split + one sum launch) overwrites dw: NaN-filled target,
        # NaN-filled workspace, called twice; where it does not apply the size query says so
        lib().dtt_conv3x3_wgrad_ws_bytes.restype = C.c_int64
        need = lib().dtt_conv3x3_wgrad_ws_bytes(Bn, Fq, T, Ci, Co)
        assert (need > 0) == ((Ci == 32 and Co == 32) or (Ci % 64 == 0 and Co % 64 == 0))
        if need > 0:
            dw2 = torch.full((Co, 9, Ci), float("nan"), dtype=torch.float32, device=dev)
            ws = torch.full((need // 4,), float("nan"), dtype=torch.float32, device=dev)
            for _ in range(2):
                check(lib().dtt_conv3x3_wgrad_ws(ptr(_nhwc(dy).to(dev)), ptr(xd), ptr(dw2), ptr(ws), C.c_int64(need), Bn, Fq, T,
                                                 Ci, Co, 1, stream_ptr()), "wgrad_ws")
            assert rel_err(dw2, wr.grad.permute(0, 2, 3, 1).reshape(Co, 9, Ci)) < TOL[dt]
            assert rel_err(dw2, dwp) < 1e-5
            assert lib().dtt_conv3x3_wgrad_ws(ptr(_nhwc(dy).to(dev)), ptr(xd), ptr(dw2), ptr(ws), C.c_int64(need - 16), Bn, Fq,
                                              T, Ci, Co, 1, stream_ptr()) != 0


def test_conv3x3_wgrad_workspace_many_splits():
    """The workspace weight gradient at shapes where every CU gets its own split (64 -> 64 at 16 x 257 x 63: 256
    slabs; 512 -> 512: 4 slabs x 64 channel-tile pairs; 32 -> 32: 768 slabs) against the atomic form."""
    import ctypes as C
    from dtt_amd._lib import lib, check, ptr, stream_ptr
    lib().dtt_conv3x3_wgrad_ws_bytes.restype = C.c_int64
    g = torch.Generator().manual_seed(11)
    for (Bn, Fq, T, Cc) in [(16, 257, 63, 64), (4, 65, 16, 512), (4, 513, 126, 32)]:
        x = torch.randn(Bn, Fq, T, Cc, generator=g).to(torch.bfloat16).cuda()
        dy = torch.randn(Bn, Fq, T, Cc, generator=g).to(torch.bfloat16).cuda()
        ref = torch.zeros(Cc, 9, Cc, device="cuda")
        check(lib().dtt_conv3x3_wgrad(ptr(dy), ptr(x), ptr(ref), Bn, Fq, T, Cc, Cc, 1, stream_ptr()), "wgrad")
        need = lib().dtt_conv3x3_wgrad_ws_bytes(Bn, Fq, T, Cc, Cc)
        assert need > 0
        ws = torch.full((need // 4,), float("nan"), device="cuda")
        dw = torch.full((Cc, 9, Cc), float("nan"), device="cuda")
        check(lib().dtt_conv3x3_wgrad_ws(ptr(dy), ptr(x), ptr(dw), ptr(ws), C.c_int64(need), Bn, Fq, T, Cc, Cc, 1, stream_ptr()),
              "wgrad_ws")
        assert rel_err(dw, ref) < 1e-5, (Bn, Fq, T, Cc)



@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Co,Ci", [(32, 32), (64, 24), (8, 136)])
def test_pack_conv3_dgrad_is_the_rotated_transpose(dt, Co, Ci):
    """One-launch pack of the data-gradient conv's weight: bit-exact against flip + permute + cast."""
    from dtt_amd import ops
    torch.manual_seed(5)
    w = torch.randn(Co, Ci, 3, 3, device="cuda")
    got = ops._pack_dgrad(w, "s1", dt)
    ref = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Ci, 9 * Co).to(dt)
    assert got.shape == ref.shape and got.dtype == dt
    assert torch.equal(got, ref)


@pytest.mark.parametrize("M,N,K,res,f32", [(2050, 3072, 128, True, False), (4100, 3000, 64, False, True),
                                          (33280, 4096, 256, False, False)])
def test_gemm256_k_contiguous(M, N, K, res, f32):
    """The 256 x 256 x 64 LDS-DMA kernel (gemm256.hip) takes the large k-contiguous bf16 products (>= 192 tiles):
    ragged edges in both dimensions, bias, residual, bf16 and fp32 stores, against fp32 math on the same bf16
    inputs; and the same call with the kernel switched off must agree to bf16 rounding."""
    from dtt_amd._lib import gemm_raw
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = torch.randn(N, K, device=dev, generator=g).bfloat16()
    b = torch.randn(N, device=dev, generator=g)
    r = torch.randn(M, N, device=dev, generator=g).bfloat16() if res else None
    c = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    kw = dict(A=a, B=w, C=c, bias=b, M=M, N=N, K=K, a_ms=K, a_ks=1, b_ks=1, b_ns=K, c_ms=N, c_ns=1, dtype=1,
              bias_mode=1, out_f32=1 if f32 else 0)
    if res:
        kw.update(R=r, r_ms=N, r_ns=1)
    gemm_raw(**kw)
    ref = a.float() @ w.float().t() + b
    if res:
        ref = ref + r.float()
    assert rel_err(c, ref) < (1e-5 if f32 else 1e-2)
    assert float((c.float() - ref).abs().max() / ref.abs().max()) < (1e-5 if f32 else 1e-2)


@pytest.mark.parametrize("R,H,L", [(33280, 512, 16), (8320, 512, 65)])
def test_gemm256_weight_gradient_with_row_shift(R, H, L):
    """Row-contiguous split-K form of gemm256.hip: dW_hh = dG^T h_{t-1} with the LSTM row shift (zero rows at the
    sequence edge), fp32 atomic accumulation, against fp32 math."""
    from dtt_amd._lib import gemm_raw
    dev = "cuda"
    G4 = 4 * H
    g = torch.Generator(device=dev).manual_seed(R + H)
    dG = torch.randn(R, 2, G4, device=dev, generator=g).bfloat16()
    y = torch.randn(R, 2 * H, device=dev, generator=g).bfloat16()
    k = torch.arange(R, device=dev)
    for di, (shift, edge) in enumerate(((-2, 0), (2, L - 1))):
        out = torch.zeros(G4, H, device=dev)
        gemm_raw(A=dG[:, di], B=y[:, di * H:], C=out, M=G4, N=H, K=R, splitk=8, a_ms=1, a_ks=2 * G4, b_ks=2 * H, b_ns=1,
                 c_ms=H, c_ns=1, dtype=1, atomic=1, out_f32=1, b_mode=2, sh_shift=shift, sh_heads=2, sh_L=L, sh_edge=edge)
        keep = (k // 2) % L != edge
        ys = torch.zeros(R, H, device=dev)
        ys[keep] = y[(k + shift).clamp(0, R - 1)[keep], di * H:(di + 1) * H].float()
        ref = dG[:, di].float().t() @ ys
        assert rel_err(out, ref) < 1e-4
