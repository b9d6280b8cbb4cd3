"""Direct check of the cluster LSTM kernels against the streaming kernels on the same inputs (C ABI)."""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-separation_amd"))
from dtt_amd._lib import lib, check, ptr, stream_ptr, dtype_code
from dtt_amd import ops
dev = "cuda"

def run(N, L, heads, H, dt, seed=0):
    g = torch.Generator().manual_seed(seed)
    R = N * L
    d = dtype_code(dt)
    xg = (torch.randn(R, 8 * H, generator=g) * 0.7).to(dev).to(dt)
    whh = (torch.randn(2, 4 * H, H, generator=g) / H ** 0.5).to(dev)
    dy = torch.randn(R, 2 * H, generator=g).to(dev).to(dt)
    # reference: streaming kernels (row-major activations)
    wp = torch.stack([ops._pack_whh_fwd(whh[0], dt), ops._pack_whh_fwd(whh[1], dt)]).contiguous()
    y0 = torch.empty(R, 2 * H, device=dev, dtype=dt); g0 = torch.empty(R, 2, 4 * H, device=dev, dtype=dt)
    c0 = torch.empty(R, 2, H, device=dev)
    check(lib().dtt_lstm_fwd(ptr(xg), ptr(wp), ptr(y0), ptr(g0), ptr(c0), N, L, heads, H, d, stream_ptr()), "fwd")
    wpt = torch.stack([ops._pack_whh_bwd(whh[0], dt), ops._pack_whh_bwd(whh[1], dt)]).contiguous()
    dG0 = torch.zeros(R, 2, 4 * H, device=dev, dtype=dt)
    check(lib().dtt_lstm_bwd(ptr(dy), ptr(g0), ptr(c0), ptr(wpt), ptr(dG0), N, L, heads, H, d, stream_ptr()), "bwd")
    # cluster kernels
    geom = ops._cluster_geometry(N, H, d)
    NT = (N + 31) // 32
    wt = whh.to(dt).contiguous()
    y1 = torch.full((R, 2 * H), float("nan"), device=dev, dtype=dt)
    g1 = torch.empty(2, NT, L, 4 * H * 32, device=dev, dtype=dt); c1 = torch.empty(2, NT, L, H * 32, device=dev)
    sync = torch.empty(lib().dtt_lstm_cluster_sync_words(geom[2]), device=dev, dtype=torch.int32)
    check(lib().dtt_lstm_cluster_fwd(ptr(xg), ptr(wt), ptr(y1), ptr(g1), ptr(c1), ptr(sync), N, L, heads, H, d, stream_ptr()), "cfwd")
    torch.cuda.synchronize()
    st_f = int(sync[-64])
    whhT = whh.transpose(1, 2).to(dt).contiguous()
    dG1 = torch.full((R, 2, 4 * H), float("nan"), device=dev, dtype=dt)
    dcb = torch.empty(2, NT, H * 32, device=dev); db = torch.zeros(2 * 4 * H, device=dev)
    sync2 = torch.empty_like(sync)
    check(lib().dtt_lstm_cluster_bwd(ptr(dy), ptr(g1), ptr(c1), ptr(whhT), ptr(dG1), ptr(dcb), ptr(db), ptr(sync2), N, L, heads, H, d, stream_ptr()), "cbwd")
    torch.cuda.synchronize()
    st_b = int(sync2[-64])
    ey = float((y1.float() - y0.float()).norm() / y0.float().norm())
    eg = float((dG1.float() - dG0.float()).norm() / dG0.float().norm())
    eb = float((db - dG0.float().sum(0).reshape(-1)).norm() / dG0.float().sum(0).norm())
    print(f"N={N} L={L} H={H} {str(dt)[6:]} geom(U,G,S)={geom} status f/b={st_f}/{st_b} err y={ey:.2e} dG={eg:.2e} dbias={eb:.2e}", flush=True)
    return ey, eg, eb

def bench(N, L, heads, H, dt):
    d = dtype_code(dt); R = N * L
    xg = (torch.randn(R, 8 * H, device=dev) * 0.7).to(dt); whh = torch.randn(2, 4 * H, H, device=dev) / H ** 0.5
    dy = torch.randn(R, 2 * H, device=dev).to(dt)
    geom = ops._cluster_geometry(N, H, d); NT = (N + 31) // 32
    wt = whh.to(dt).contiguous(); whhT = whh.transpose(1, 2).to(dt).contiguous()
    y1 = torch.empty(R, 2 * H, device=dev, dtype=dt)
    g1 = torch.empty(2, NT, L, 4 * H * 32, device=dev, dtype=dt); c1 = torch.empty(2, NT, L, H * 32, device=dev)
    sync = torch.empty(lib().dtt_lstm_cluster_sync_words(geom[2]), device=dev, dtype=torch.int32)
    dG1 = torch.empty(R, 2, 4 * H, device=dev, dtype=dt); dcb = torch.empty(2, NT, H * 32, device=dev); db = torch.zeros(2 * 4 * H, device=dev)
    def f(): check(lib().dtt_lstm_cluster_fwd(ptr(xg), ptr(wt), ptr(y1), ptr(g1), ptr(c1), ptr(sync), N, L, heads, H, d, stream_ptr()), "cfwd")
    def b(): check(lib().dtt_lstm_cluster_bwd(ptr(dy), ptr(g1), ptr(c1), ptr(whhT), ptr(dG1), ptr(dcb), ptr(db), ptr(sync), N, L, heads, H, d, stream_ptr()), "cbwd")
    out = []
    for fn in (f, b):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * N * L * 2 * H * 4 * H
    print(f"bench N={N} L={L} H={H} geom={geom}: fwd {out[0]:.0f} us ({fl / out[0] / 1e6:.0f} TF/s)  bwd {out[1]:.0f} us ({fl / out[1] / 1e6:.0f} TF/s)  status {int(sync[-64])}", flush=True)

if __name__ == "__main__":
    bad = 0
    for (N, L, heads, H, dt) in [(6, 5, 2, 64, torch.float32), (70, 7, 2, 64, torch.bfloat16), (40, 9, 2, 128, torch.bfloat16),
                                 (130, 6, 2, 256, torch.bfloat16), (520, 16, 2, 512, torch.bfloat16), (64, 33, 2, 512, torch.bfloat16),
                                 (66, 5, 2, 512, torch.float32), (96, 4, 2, 1024, torch.bfloat16)]:
        ey, eg, eb = run(N, L, heads, H, dt)
        tol = 2e-5 if dt == torch.float32 else 3e-2
        bad += not (ey < tol and eg < 2 * tol and eb < 2 * tol)
    if bad == 0 and len(sys.argv) > 1:
        bench(2080, 16, 2, 512, torch.bfloat16)
        bench(512, 65, 2, 512, torch.bfloat16)
        bench(8320, 16, 2, 256, torch.bfloat16)
        bench(2048, 65, 2, 256, torch.bfloat16)
        bench(2080, 47, 2, 1024, torch.bfloat16)
    print("FAILED" if bad else "ALL OK")
    sys.exit(1 if bad else 0)
