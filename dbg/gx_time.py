import os, sys, torch
sys.path.insert(0, "speech-separation_amd")
from dtt_amd._lib import gemm_raw
dev, bf = "cuda", torch.bfloat16
def timeit(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in ((8192, 8192, 4096), (33280, 4096, 256), (33280, 512, 2048)):
    a = torch.randn(M, K, device=dev).to(bf); b = torch.randn(N, K, device=dev).to(bf); c = torch.empty(M, N, device=dev, dtype=bf)
    t = timeit(lambda: gemm_raw(A=a, B=b, C=c, M=M, N=N, K=K, a_ms=K, a_ks=1, b_ks=1, b_ns=K, c_ms=N, c_ns=1, dtype=1))
    print(f"{os.environ.get('DTT_HIP_LIB','default')[-12:]} {M}x{N}x{K}: {t:7.1f} us {2.0*M*N*K/t/1e6:6.0f} TF/s", flush=True)
