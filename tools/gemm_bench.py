#!/usr/bin/env python3
"""Timing of the LSTM-adjacent products of a cfg4 step through dtt_gemm (latent.py:87-96 and their gradients):
xg = xn W_ih^T, the output Linear, dxn = dG W_ih, dy = dout W_fc, and the weight gradients.  Kernel selection is by
environment (read once per process): DTT_GEMM256=0, DTT_GEMMX=0, DTT_GEMMX_TILE=0..4."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd"))
from dtt_amd._lib import gemm_raw  # noqa: E402

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


R, H, Cp = 33280, 512, 256
G4 = 4 * H
print({k: v for k, v in os.environ.items() if k.startswith("DTT_")})
dG = torch.randn(R, 2, G4, device=dev).to(bf)
y = torch.randn(R, 2 * H, device=dev).to(bf)
xn = torch.randn(R, Cp, device=dev).to(bf)
wih = torch.randn(2 * G4, Cp, device=dev).to(bf)
fcw = torch.randn(Cp, 2 * H, device=dev).to(bf)
bias = torch.zeros(8 * H, device=dev)
xg = torch.empty(R, 8 * H, device=dev, dtype=bf)
t = timeit(lambda: gemm_raw(A=xn, B=wih, C=xg, bias=bias, M=R, N=8 * H, K=Cp, a_ms=Cp, a_ks=1, b_ks=1, b_ns=Cp, c_ms=8 * H, c_ns=1, dtype=1, bias_mode=1))
print(f"xg   [33280 x 4096 x 256]  A x B^T : {t:6.1f} us  C write {R * 8 * H * 2 / t / 1e3:5.0f} GB/s")
out = torch.empty(R, Cp, device=dev, dtype=bf)
res = torch.randn(R, Cp, device=dev).to(bf)
t = timeit(lambda: gemm_raw(A=y, B=fcw, C=out, bias=bias[:Cp], R=res, M=R, N=Cp, K=2 * H, a_ms=2 * H, a_ks=1, b_ks=1, b_ns=2 * H, c_ms=Cp, c_ns=1, r_ms=Cp, r_ns=1, dtype=1, bias_mode=1))
print(f"fc   [33280 x 256 x 1024]  A x B^T : {t:6.1f} us  {2.0 * R * Cp * 2 * H / t / 1e6:5.0f} TF/s")
dxn = torch.empty(R, Cp, device=dev, dtype=bf)
t = timeit(lambda: gemm_raw(A=dG, B=wih, C=dxn, M=R, N=Cp, K=2 * G4, a_ms=2 * G4, a_ks=1, b_ks=Cp, b_ns=1, c_ms=Cp, c_ns=1, dtype=1))
print(f"dxn  [33280 x 256 x 4096]  A x B   : {t:6.1f} us  A read {R * 2 * G4 * 2 / t / 1e3:5.0f} GB/s  {2.0 * R * Cp * 2 * G4 / t / 1e6:5.0f} TF/s")
dy = torch.empty(R, 2 * H, device=dev, dtype=bf)
t = timeit(lambda: gemm_raw(A=out, B=fcw, C=dy, M=R, N=2 * H, K=Cp, a_ms=Cp, a_ks=1, b_ks=2 * H, b_ns=1, c_ms=2 * H, c_ns=1, dtype=1))
print(f"dy   [33280 x 1024 x 256]  A x B   : {t:6.1f} us  C write {R * 2 * H * 2 / t / 1e3:5.0f} GB/s")
