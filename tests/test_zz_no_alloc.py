"""dtt_sisnr_noalloc (the one entry point that allocates nothing: its scratch is a static device array)
against dtt_sisnr through the C ABI, bit for bit, on two calls in a row, the second with a smaller input
(the array keeps the first call's contents)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _both(E, S, B, L, seed):
    from dtt_amd._lib import lib, check, ptr, stream_ptr
    g = torch.Generator().manual_seed(seed)
    est = torch.randn(E, B, L, generator=g).to(DEV)
    srcs = [torch.randn(B, L, generator=g).to(DEV) for _ in range(S)]
    arr = (C.c_void_p * S)(*[t.data_ptr() for t in srcs])
    nch = max(1, min(32, L // 2048))
    part = torch.empty(E * S * B * nch * 2, device=DEV)
    alpha = torch.empty(E * S * B, device=DEV)
    ref = torch.empty(E, S, B, device=DEV)
    check(lib().dtt_sisnr(ptr(est), arr, ptr(part), ptr(alpha), ptr(ref), E, S, B, C.c_int64(L), nch, C.c_float(1e-5),
                          stream_ptr()), "dtt_sisnr")
    out = torch.full((E, S, B), float("nan"), device=DEV)
    check(lib().dtt_sisnr_noalloc(ptr(est), arr, ptr(out), E, S, B, C.c_int64(L), C.c_float(1e-5), stream_ptr()),
          "dtt_sisnr_noalloc")
    torch.cuda.synchronize()
    return ref, out


def test_noalloc_sisnr_matches_dtt_sisnr_twice_and_rejects_oversize():
    from dtt_amd._lib import lib, ptr, stream_ptr
    ref, out = _both(3, 2, 16, 32000, seed=1)                 # SI-SNRi at the bench size: preds + mixture vs 2 sources
    assert torch.equal(ref, out) and torch.isfinite(out).all()
    ref, out = _both(2, 2, 3, 5000, seed=2)                   # smaller, right behind: stale scratch must not matter
    assert torch.equal(ref, out) and torch.isfinite(out).all()
    ref, out = _both(1, 1, 1, 700, seed=3)                    # one chunk
    assert torch.equal(ref, out)
    # a size beyond the static array is refused with a clear error (and nothing is launched)
    L_ = lib()
    cap = L_.dtt_sisnr_noalloc_max_floats()
    B = cap // (2 * 2 * (2 * 32 + 1)) + 1
    est = torch.zeros(2, B, 65536, device=DEV)
    src = [torch.zeros(B, 65536, device=DEV) for _ in range(2)]
    arr = (C.c_void_p * 2)(*[t.data_ptr() for t in src])
    o = torch.zeros(2, 2, B, device=DEV)
    rc = L_.dtt_sisnr_noalloc(ptr(est), arr, ptr(o), 2, 2, B, C.c_int64(65536), C.c_float(1e-5), stream_ptr())
    assert rc != 0 and b"exceed the static" in L_.dtt_last_error()
