#!/usr/bin/env python3
"""dxn = dG W_ih (33 280 x 256 x 4096; latent.py:87-92 backward) with W_ih as stored ([4096][256], A x B) and
transposed ([256][4096], A x B^T: the k-contiguous tiles of gemm256 / gemmx).  Kernel selection by environment."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from dtt_amd._lib import gemm_raw, lib  # noqa: E402
dev, bf = "cuda", torch.bfloat16


def timeit(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


R, K, N = 33280, 4096, 256
dG = (torch.randn(R, K, device=dev) * 0.1).to(bf)
w = (torch.randn(K, N, device=dev) * 0.1).to(bf)
wT = w.t().contiguous()
o1 = torch.empty(R, N, device=dev, dtype=bf)
o2 = torch.empty(R, N, device=dev, dtype=bf)
t1 = timeit(lambda: gemm_raw(A=dG, B=w, C=o1, M=R, N=N, K=K, a_ms=K, a_ks=1, b_ks=N, b_ns=1, c_ms=N, c_ns=1, dtype=1))
t2 = timeit(lambda: gemm_raw(A=dG, B=wT, C=o2, M=R, N=N, K=K, a_ms=K, a_ks=1, b_ks=1, b_ns=K, c_ms=N, c_ns=1, dtype=1))
err = float((o1.float() - o2.float()).abs().max() / o1.float().abs().max())
env = {k: v for k, v in os.environ.items() if k.startswith("DTT_")}
print(f"{env}: A x B {t1:6.1f} us   A x B^T {t2:6.1f} us   gemmx launches {lib().dtt_gemmx_launches()}  rel diff {err:.1e}")
