"""Direct check of the register-stationary LSTM kernels against the streaming kernels on the same inputs (C ABI)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-separation_amd"))
from dtt_amd._lib import lib, check, ptr, stream_ptr, dtype_code
from dtt_amd import ops
dev = "cuda"


def granules_to_rows(t, W, N, L, heads, H):
    """[2][NT][L][H/4][32][W/H][4] granule layout -> row-major [R][2][W] (W = 4H gates or H cell state)."""
    NT = t.shape[1]
    g = W // H
    v = t.view(2, NT, L, H // 4, 32, g, 4).permute(1, 4, 2, 0, 5, 3, 6).reshape(NT * 32, L, 2, W)[:N]   # [n][L][2][g*H]
    v = v.view(N // heads, heads, L, 2, W).permute(0, 2, 1, 3, 4).reshape(N * L, 2, W)
    return v.contiguous()


def run(N, L, heads, H, seed=0, bwd=True, rounds=2, load=True):
    """`rounds` launches with DIFFERENT inputs into the SAME buffers (a stale read of the hand-off would return the
    previous round's bytes), every element checked (max abs, not only the norm), optionally beside a stream of large
    copies on another HIP stream (uneven load on the memory system while the clusters exchange their tiles)."""
    dt = torch.bfloat16
    R = N * L
    d = dtype_code(dt)
    NT = (N + 31) // 32
    y1 = torch.full((R, 2 * H), float("nan"), device=dev, dtype=dt)
    g1 = torch.zeros(2, NT, L, 4 * H * 32, device=dev, dtype=dt); c1 = torch.zeros(2, NT, L, H * 32, device=dev)
    dG1 = torch.full((R, 2, 4 * H), float("nan"), device=dev, dtype=dt)
    dcb = torch.empty(2, NT, H * 32, device=dev)
    nsync = lib().dtt_lstm_rs_sync_words(N, H)
    assert nsync > 0, "shape not covered"
    sync = torch.empty(nsync, device=dev, dtype=torch.int32); sync2 = torch.empty_like(sync)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device=dev, dtype=torch.float32) if load else None
    ok = True
    for rnd in range(rounds):
        g = torch.Generator().manual_seed(seed + 17 * rnd)
        xg = (torch.randn(R, 8 * H, generator=g) * 0.7).to(dev).to(dt)
        whh = (torch.randn(2, 4 * H, H, generator=g) / H ** 0.5).to(dev)
        dy = torch.randn(R, 2 * H, generator=g).to(dev).to(dt)
        wp = torch.stack([ops._pack_whh_fwd(whh[0], dt), ops._pack_whh_fwd(whh[1], dt)]).contiguous()
        y0 = torch.empty(R, 2 * H, device=dev, dtype=dt); g0 = torch.empty(R, 2, 4 * H, device=dev, dtype=dt)
        c0 = torch.empty(R, 2, H, device=dev)
        check(lib().dtt_lstm_fwd(ptr(xg), ptr(wp), ptr(y0), ptr(g0), ptr(c0), N, L, heads, H, d, stream_ptr()), "fwd")
        wt = whh.to(dt).contiguous()
        torch.cuda.synchronize()
        if load:
            with torch.cuda.stream(side):
                for _ in range(12):
                    junk[: 32 << 20].copy_(junk[32 << 20:])
        check(lib().dtt_lstm_rs_fwd(ptr(xg), ptr(wt), ptr(y1), ptr(g1), ptr(c1), ptr(sync), N, L, heads, H, d, stream_ptr()), "rs fwd")
        torch.cuda.synchronize()
        st_f = int(sync[0])
        ey = float((y1.float() - y0.float()).norm() / y0.float().norm())
        my = float((y1.float() - y0.float()).abs().max())
        # per hidden unit (column): one wrong wave = 8 .. 32 columns of one cluster member cannot hide inside a norm
        cy = float(((y1.float() - y0.float()).norm(dim=0) / (y0.float().norm(dim=0) + 1e-12)).max())
        egt = float((granules_to_rows(g1, 4 * H, N, L, heads, H).float() - g0.float()).norm() / g0.float().norm())
        ec = float((granules_to_rows(c1, H, N, L, heads, H) - c0).norm() / c0.norm())
        msg = f"N={N} L={L} H={H} round {rnd} status f={st_f} err y={ey:.2e} (max {my:.3f}) gates={egt:.2e} c={ec:.2e}"
        msg += f" worst-unit y={cy:.2e}"
        ok = ok and st_f == 0 and ey < 3e-2 and egt < 3e-2 and ec < 3e-2 and my < 0.06 and cy < 6e-2
        if bwd and hasattr(lib(), "dtt_lstm_rs_bwd"):
            wpt = torch.stack([ops._pack_whh_bwd(whh[0], dt), ops._pack_whh_bwd(whh[1], dt)]).contiguous()
            dG0 = torch.zeros(R, 2, 4 * H, device=dev, dtype=dt)
            check(lib().dtt_lstm_bwd(ptr(dy), ptr(g0), ptr(c0), ptr(wpt), ptr(dG0), N, L, heads, H, d, stream_ptr()), "bwd")
            whhT = whh.transpose(1, 2).to(dt).contiguous()
            db = torch.zeros(2 * 4 * H, device=dev)
            torch.cuda.synchronize()
            if load:
                with torch.cuda.stream(side):
                    for _ in range(12):
                        junk[: 32 << 20].copy_(junk[32 << 20:])
            check(lib().dtt_lstm_rs_bwd(ptr(dy), ptr(g1), ptr(c1), ptr(whhT), ptr(dG1), ptr(dcb), ptr(db), ptr(sync2), N, L, heads, H, d, stream_ptr()), "rs bwd")
            torch.cuda.synchronize()
            st_b = int(sync2[0])
            eg = float((dG1.float() - dG0.float()).norm() / dG0.float().norm())
            mg = float((dG1.float() - dG0.float()).abs().max() / dG0.float().abs().max())
            d2 = (dG1.float() - dG0.float()).reshape(R, -1)
            cg = float((d2.norm(dim=0) / (dG0.float().reshape(R, -1).norm(dim=0) + 1e-12)).max())
            eb = float((db - dG0.float().sum(0).reshape(-1)).norm() / dG0.float().sum(0).norm())
            msg += f" | status b={st_b} dG={eg:.2e} (max {mg:.3f} of the largest) dbias={eb:.2e}"
            msg += f" worst-column dG={cg:.2e}"
            ok = ok and st_b == 0 and eg < 6e-2 and eb < 6e-2 and mg < 0.05 and cg < 0.12
        print(msg, "OK" if ok else "BAD", flush=True)
    return ok


def bench(N, L, heads, H):
    dt = torch.bfloat16
    d = dtype_code(dt); R = N * L
    xg = (torch.randn(R, 8 * H, device=dev) * 0.7).to(dt); whh = torch.randn(2, 4 * H, H, device=dev) / H ** 0.5
    dy = torch.randn(R, 2 * H, device=dev).to(dt)
    NT = (N + 31) // 32
    wt = whh.to(dt).contiguous(); whhT = whh.transpose(1, 2).to(dt).contiguous()
    y1 = torch.empty(R, 2 * H, device=dev, dtype=dt)
    g1 = torch.empty(2, NT, L, 4 * H * 32, device=dev, dtype=dt); c1 = torch.empty(2, NT, L, H * 32, device=dev)
    sync = torch.empty(lib().dtt_lstm_rs_sync_words(N, H), device=dev, dtype=torch.int32)
    dG1 = torch.empty(R, 2, 4 * H, device=dev, dtype=dt); dcb = torch.empty(2, NT, H * 32, device=dev); db = torch.zeros(2 * 4 * H, device=dev)
    def f(): check(lib().dtt_lstm_rs_fwd(ptr(xg), ptr(wt), ptr(y1), ptr(g1), ptr(c1), ptr(sync), N, L, heads, H, d, stream_ptr()), "rs fwd")
    def fi(): check(lib().dtt_lstm_rs_fwd(ptr(xg), ptr(wt), ptr(y1), None, ptr(c1), ptr(sync), N, L, heads, H, d, stream_ptr()), "rs fwd")
    def b(): check(lib().dtt_lstm_rs_bwd(ptr(dy), ptr(g1), ptr(c1), ptr(whhT), ptr(dG1), ptr(dcb), ptr(db), ptr(sync), N, L, heads, H, d, stream_ptr()), "rs bwd")
    fns = [("fwd", f), ("fwd-nosave", fi)] + ([("bwd", b)] if hasattr(lib(), "dtt_lstm_rs_bwd") else [])
    fl = 2.0 * N * L * 2 * H * 4 * H
    out = []
    for name, fn in fns:
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        out.append(f"{name} {us:.0f} us ({fl / us / 1e6:.0f} TF/s)")
    print(f"bench N={N} L={L} H={H}: " + "  ".join(out) + f"  status {int(sync[0])}", flush=True)


if __name__ == "__main__":  # noqa
    ok = True
    for (N, L, heads, H) in [(64, 3, 2, 512), (70, 7, 2, 512), (520, 16, 2, 512), (64, 33, 2, 512), (2080, 16, 2, 512), (1024, 5, 2, 512), (1536, 6, 4, 512), (1000, 4, 1, 512),
                             (130, 6, 2, 256), (8320, 16, 2, 256), (2048, 5, 2, 256)]:
        ok = run(N, L, heads, H) and ok
    if ok and len(sys.argv) > 1:
        bench(2080, 16, 2, 512)
        bench(512, 65, 2, 512)
        bench(8320, 16, 2, 256)
        bench(2048, 65, 2, 256)
    print("ALL OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)
