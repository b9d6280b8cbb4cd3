#!/usr/bin/env python3
"""Per-shape timing of the 3x3 stride-1 convolution paths: the 16x16-pixel tile kernel (dtt_conv3x3_s1, with and
without its fused InstanceNorm statistics) against the implicit GEMM through dtt_gemm (gemmx.hip with DTT_GEMMX=1,
gemm_fast_kernel with DTT_GEMMX=0: the switch is read once per process)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd"))
from dtt_amd import ops  # noqa: E402

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


SHAPES = [(16, 257, 63, 64), (16, 129, 32, 128), (16, 65, 16, 512), (16, 65, 16, 256), (1600, 22, 22, 64),
          (1600, 11, 11, 128), (1600, 6, 6, 256), (1600, 3, 3, 512), (64, 257, 63, 64), (64, 129, 32, 128), (64, 65, 16, 256)]
print("DTT_GEMMX =", os.environ.get("DTT_GEMMX", "1"))
for (B, Fq, T, Cc) in SHAPES:
    x = torch.randn(B, Fq, T, Cc, device=dev).to(bf)
    w = (torch.randn(Cc, 9 * Cc, device=dev) * 0.05).to(bf)
    bias = torch.randn(Cc, device=dev)
    out = torch.empty_like(x)
    geom = dict(cB=B, cFi=Fq, cTi=T, cCi=Cc, cFo=Fq, cTo=T, cstride=1, ctrans=0)
    fl = 2.0 * B * Fq * T * 9 * Cc * Cc
    t_tile = timeit(lambda: ops.conv3x3_s1(x, w, bias))
    t_stat = timeit(lambda: ops.conv3x3_s1(x, w, bias, want_stats=True))
    t_gemm = timeit(lambda: ops.conv_gemm(x, w, bias, out, geom))
    t_ins = timeit(lambda: ops.in_stats(out))
    print(f"B={B:5d} {Fq:3d}x{T:<3d} C={Cc:3d}: tile {t_tile:6.1f} us ({fl / t_tile / 1e6:5.0f} TF/s)  tile+stats {t_stat:6.1f}  "
          f"dtt_gemm {t_gemm:6.1f} us ({fl / t_gemm / 1e6:5.0f} TF/s)  in_stats {t_ins:5.1f}", flush=True)
