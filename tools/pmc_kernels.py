"""Runs the roofline kernels of bench.py a few times so that rocprofv3 --pmc can count their HBM traffic:
the register-stationary recurrent kernel (lstm_rs.hip) at the cfg4 time-axis shape (N = 2080, L = 16, H = 512), and the two 3x3
convs (64->64 at [B,257,63], 32->32 at [B,513,126]) at B = 16 (cfg4) and B = 64 (cfg2).
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_rd -- python3 tools/pmc_kernels.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_wr -- python3 tools/pmc_kernels.py
(separate passes: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass); then tools/pmc_parse.py."""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "speech-separation_amd"))
from dtt_amd import ops  # noqa: E402
from dtt_amd._lib import lib, check, ptr, stream_ptr, dtype_code  # noqa: E402

dev = "cuda"
dt = torch.bfloat16
d = dtype_code(dt)
N, L, heads, H = 2080, 16, 2, 512
R = N * L
xg = (torch.randn(R, 8 * H, device=dev) * 0.5).to(dt)
whh = (torch.randn(2, 4 * H, H, device=dev) / H ** 0.5).to(dt).contiguous()
y = torch.empty(R, 2 * H, device=dev, dtype=dt)
NT = (N + 31) // 32
gates = torch.empty(2, NT, L, 4 * H * 32, device=dev, dtype=dt)
cst = torch.empty(2, NT, L, H * 32, device=dev, dtype=torch.float32)
sync = torch.empty(lib().dtt_lstm_rs_sync_words(N, H), device=dev, dtype=torch.int32)
for _ in range(5):
    check(lib().dtt_lstm_rs_fwd(ptr(xg), ptr(whh), ptr(y), ptr(gates), ptr(cst), ptr(sync), N, L, heads, H, d,
                                stream_ptr()), "dtt_lstm_rs_fwd")
torch.cuda.synchronize()
for B in (16, 64):
    for (Fq, T, Cc) in [(257, 63, 64), (513, 126, 32)]:
        x = torch.randn(B, Fq, T, Cc, device=dev).to(dt)
        w = torch.randn(Cc, Cc, 3, 3, device=dev) / (3.0 * Cc ** 0.5)
        wp = ops._pack_fwd(w, "s1", dt)
        bias = torch.zeros(Cc, device=dev)
        for _ in range(5):
            ops.conv3x3_s1(x, wp, bias, want_stats=True)
        torch.cuda.synchronize()
