"""dtt_multi_pack (pack.hip): the one launch that derives a step's packed weight tensors (bf16 casts, tap-major conv
weights, transposed LSTM / Linear / TDF weights; the layouts PyTorch's own modules keep in
src/model/DTTNet/blocks/block_tfc_tdf.py:12-32 and modules/latent.py:87-96) must equal torch's strided copy + cast
bit for bit on each of its three walks: vector casts, tiles through LDS, element-wise."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(pairs):
    from dtt_amd import ops
    from dtt_amd._lib import lib, check, ptr, stream_ptr
    pl = ops._PackPlan()
    pl.pairs = pairs
    pl.build()
    descs, chunks, n = pl.table
    check(lib().dtt_multi_pack(ptr(descs), ptr(chunks), n, stream_ptr()), "dtt_multi_pack")
    torch.cuda.synchronize()
    return n


def _cases():
    g = torch.Generator(device="cpu").manual_seed(5)

    def r(*shape, dtype=torch.float32):
        return torch.randn(*shape, generator=g).to(DEV).to(dtype)
    out = []
    # contiguous casts: vector walk with a ragged tail, fp32 -> fp32, a source off the 16-byte grid (element walk)
    out.append(("cast_tail", r(3 * 8192 + 5), None, torch.bfloat16))
    out.append(("copy_fp32", r(8192 + 3), None, torch.float32))
    out.append(("cast_unaligned", r(5000)[1:], None, torch.bfloat16))
    out.append(("cast_2d_slice", r(40, 96)[:, :80], None, torch.bfloat16))
    # 2-D transposes (LSTM / Linear weights), ragged against the 64 x 64 tile
    out.append(("t_300x77", r(300, 77).t(), None, torch.bfloat16))
    out.append(("t_2048x512", r(2048, 512).t(), None, torch.bfloat16))
    out.append(("t_3x1000", r(3, 1000).t(), None, torch.float32))
    out.append(("t_1000x3", r(1000, 3).t(), None, torch.bfloat16))
    out.append(("t_bf16_src", r(130, 257, dtype=torch.bfloat16).t(), None, torch.float32))
    # conv weights [Co][Ci][3][3] -> tap-major [Co][3][3][Ci] (forward) and [Ci][3][3][Co] (data gradient)
    out.append(("conv_fwd", r(64, 128, 3, 3).permute(0, 2, 3, 1), None, torch.bfloat16))
    out.append(("conv_dgrad", r(96, 40, 3, 3).permute(1, 2, 3, 0), None, torch.bfloat16))
    out.append(("conv_small", r(5, 7, 3, 3).permute(1, 2, 3, 0), None, torch.bfloat16))
    # data-gradient pack of a stride-1 conv: taps walked backwards (stride -1) as ONE descriptor
    from dtt_amd import ops
    for co, ci in ((96, 40), (5, 7)):
        w = r(co, ci, 3, 3)
        src = ops._StridedSource(w, (ci, 3, 3, co), (9, -3, -1, 9 * ci), 8, None)
        out.append(("conv_dgrad_s1_%d" % co, src, None, torch.bfloat16, w.flip(2, 3).permute(1, 2, 3, 0)))
    # two batch dims around the transposed pair
    out.append(("batch2", r(2, 3, 40, 9).permute(0, 1, 3, 2), None, torch.bfloat16))
    # destination is a slice of a larger buffer (stacked LSTM directions)
    big = torch.zeros(2, 70, 200, dtype=torch.bfloat16, device=DEV)
    out.append(("stacked", r(200, 70).t(), big[1], torch.bfloat16))
    return out


def test_multi_pack_walks_equal_torch_copies():
    from dtt_amd import ops
    cases = _cases()
    pairs, wants = [], []
    for name, v, dst, dt, *real in cases:
        d = torch.full(v.shape, 7.0, dtype=dt, device=DEV) if dst is None else dst
        want = torch.empty(v.shape, dtype=dt, device=DEV)
        want.copy_(real[0] if real else v)
        pairs.append((d, v))
        wants.append((name, d, want))
    n = _run(pairs)
    assert n > len(cases)
    for name, d, want in wants:
        assert torch.equal(d, want), name
    # what each case was cut into: the tiles really are used
    from dtt_amd._lib import lib
    import ctypes as C
    modes = {}
    for (name, v, *_), (d, _) in zip(cases, pairs):
        dims = ops._merge_dims(v.shape, v.stride(), d.stride())
        modes[name] = (len(dims), v.numel())
    assert modes["t_2048x512"][0] == 2 and modes["conv_fwd"][0] == 3


def test_multi_pack_chunk_counts_follow_the_walk():
    """dtt_multi_pack_nchunks: 8192-element chunks for the vector / element walks, TI x TJ tiles otherwise."""
    import ctypes as C
    from dtt_amd import ops
    from dtt_amd._lib import lib

    def nch(v, d):
        dims = ops._merge_dims(v.shape, v.stride(), d.stride())
        dims = [(1, 0, 0)] * (4 - len(dims)) + dims
        e = ops.DttPackDesc()
        e.src, e.dst = v.data_ptr(), d.data_ptr()
        for i, (n, a, b) in enumerate(dims):
            e.shape[i], e.sstride[i], e.dstride[i] = n, a, b
        e.numel = v.numel()
        e.src_dtype, e.dst_dtype = 0, 1
        return lib().dtt_multi_pack_nchunks(C.byref(e))
    a = torch.zeros(3 * 8192 + 5, device=DEV)
    assert nch(a, torch.empty_like(a, dtype=torch.bfloat16)) == 4
    w = torch.zeros(2048, 512, device=DEV).t()                     # 64 x 64 tiles
    assert nch(w, torch.empty(w.shape, dtype=torch.bfloat16, device=DEV)) == (2048 // 64) * (512 // 64)
    c = torch.zeros(64, 128, 3, 3, device=DEV).permute(0, 2, 3, 1)   # per Co: 9 taps x 128 channels = one tile
    assert nch(c, torch.empty(c.shape, dtype=torch.bfloat16, device=DEV)) == 64
