#!/usr/bin/env python3
"""What bounds the IDP input projection xg = xn W_ih^T (33 280 x 4096 x 256, 272 MB written)?  The same output with
K from 64 to 1024, and with the output width halved / quartered (latent.py:87-92)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from dtt_amd._lib import gemm_raw  # noqa: E402
from dxn_probe import timeit       # noqa: E402  (prints its own line first)
dev, bf = "cuda", torch.bfloat16
R = 33280
for N in (4096, 2048, 1024):
    for K in (64, 128, 256, 512, 1024):
        a = (torch.randn(R, K, device=dev) * 0.1).to(bf)
        b = (torch.randn(N, K, device=dev) * 0.1).to(bf)
        c = torch.empty(R, N, device=dev, dtype=bf)
        t = timeit(lambda: gemm_raw(A=a, B=b, C=c, M=R, N=N, K=K, a_ms=K, a_ks=1, b_ks=1, b_ns=K, c_ms=N, c_ns=1, dtype=1))
        print(f"N={N:5d} K={K:5d}: {t:7.1f} us   C write {R * N * 2 / t / 1e3:6.0f} GB/s   {2.0 * R * N * K / t / 1e6:6.0f} TF/s")
