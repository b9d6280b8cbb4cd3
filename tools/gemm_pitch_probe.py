#!/usr/bin/env python3
"""Does the 8 KB row pitch of xg / dG [33280][4096] bf16 hold the projection GEMMs back?  Same products with padded
pitches.  (latent.py:87-96)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd"))
from dtt_amd._lib import gemm_raw  # noqa: E402
from gemm_bench import timeit      # noqa: E402  (prints its own table first)
dev, bf = "cuda", torch.bfloat16
R, H, Cp = 33280, 512, 256
N8 = 8 * H
xn = torch.randn(R, Cp, device=dev).to(bf)
wih = torch.randn(N8, Cp, device=dev).to(bf)
bias = torch.zeros(N8, device=dev)
for pad in (0, 64, 128, 256, 1024):
    P = N8 + pad
    xg = torch.empty(R, P, device=dev, dtype=bf)
    t = timeit(lambda: gemm_raw(A=xn, B=wih, C=xg, bias=bias, M=R, N=N8, K=Cp, a_ms=Cp, a_ks=1, b_ks=1, b_ns=Cp, c_ms=P, c_ns=1, dtype=1, bias_mode=1))
    dG = torch.randn(R, P, device=dev).to(bf)
    dxn = torch.empty(R, Cp, device=dev, dtype=bf)
    t2 = timeit(lambda: gemm_raw(A=dG, B=wih, C=dxn, M=R, N=Cp, K=N8, a_ms=P, a_ks=1, b_ks=Cp, b_ns=1, c_ms=Cp, c_ns=1, dtype=1))
    print(f"pitch {P:5d} elements: xg {t:6.1f} us   dxn {t2:6.1f} us")
# a pure copy of the same bytes for scale
src = torch.randn(R, N8, device=dev).to(bf)
dst = torch.empty_like(src)
print(f"copy 272 MB -> 272 MB: {timeit(lambda: dst.copy_(src)):6.1f} us;  fill 272 MB: {timeit(lambda: dst.zero_()):6.1f} us")
