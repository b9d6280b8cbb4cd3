"""Runs the two roofline kernels of bench.py (the 64->64 conv at [64,257,63] and the 32->32 conv at
[64,513,126], both with fused InstanceNorm statistics) a few times so that rocprofv3 --pmc can count
their HBM traffic:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_rd -o rd -f csv -- python3 tools/pmc_kernels.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_wr -o wr -f csv -- python3 tools/pmc_kernels.py
(separate passes: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass)."""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "speech-separation_amd"))
from dtt_amd import ops  # noqa: E402

dev = "cuda"
dt = torch.bfloat16
for (B, Fq, T, Cc) in [(64, 257, 63, 64), (64, 513, 126, 32)]:
    x = torch.randn(B, Fq, T, Cc, device=dev).to(dt)
    w = torch.randn(Cc, Cc, 3, 3, device=dev) / (3.0 * Cc ** 0.5)
    wp = ops._pack_fwd(w, "s1", dt)
    bias = torch.zeros(Cc, device=dev)
    for _ in range(5):
        ops.conv3x3_s1(x, wp, bias, want_stats=True)
    torch.cuda.synchronize()
