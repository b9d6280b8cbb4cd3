"""Where does the register-stationary backward differ from the streaming kernel? (diagnostic)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-separation_amd"))
from dtt_amd._lib import lib, check, ptr, stream_ptr
from dtt_amd import ops
dev, dt = "cuda", torch.bfloat16
N, L, heads, H = (int(a) for a in sys.argv[1:5])
g = torch.Generator().manual_seed(0)
R = N * L
xg = (torch.randn(R, 8 * H, generator=g) * 0.7).to(dev).to(dt)
whh = (torch.randn(2, 4 * H, H, generator=g) / H ** 0.5).to(dev)
dy = torch.randn(R, 2 * H, generator=g).to(dev).to(dt)
wp = torch.stack([ops._pack_whh_fwd(whh[0], dt), ops._pack_whh_fwd(whh[1], dt)]).contiguous()
y0 = torch.empty(R, 2 * H, device=dev, dtype=dt); g0 = torch.empty(R, 2, 4 * H, device=dev, dtype=dt); c0 = torch.empty(R, 2, H, device=dev)
check(lib().dtt_lstm_fwd(ptr(xg), ptr(wp), ptr(y0), ptr(g0), ptr(c0), N, L, heads, H, 1, stream_ptr()), "fwd")
wpt = torch.stack([ops._pack_whh_bwd(whh[0], dt), ops._pack_whh_bwd(whh[1], dt)]).contiguous()
dG0 = torch.zeros(R, 2, 4 * H, device=dev, dtype=dt)
check(lib().dtt_lstm_bwd(ptr(dy), ptr(g0), ptr(c0), ptr(wpt), ptr(dG0), N, L, heads, H, 1, stream_ptr()), "bwd")
NT = (N + 31) // 32
wt = whh.to(dt).contiguous(); whhT = whh.transpose(1, 2).to(dt).contiguous()
y1 = torch.empty(R, 2 * H, device=dev, dtype=dt)
g1 = torch.zeros(2, NT, L, 4 * H * 32, device=dev, dtype=dt); c1 = torch.zeros(2, NT, L, H * 32, device=dev)
sync = torch.empty(lib().dtt_lstm_rs_sync_words(N, H), device=dev, dtype=torch.int32)
check(lib().dtt_lstm_rs_fwd(ptr(xg), ptr(wt), ptr(y1), ptr(g1), ptr(c1), ptr(sync), N, L, heads, H, 1, stream_ptr()), "rs fwd")
dG1 = torch.full((R, 2, 4 * H), float("nan"), device=dev, dtype=dt)
dcb = torch.empty(2, NT, H * 32, device=dev); db = torch.zeros(2 * 4 * H, device=dev)
check(lib().dtt_lstm_rs_bwd(ptr(dy), ptr(g1), ptr(c1), ptr(whhT), ptr(dG1), ptr(dcb), ptr(db), ptr(sync), N, L, heads, H, 1, stream_ptr()), "rs bwd")
torch.cuda.synchronize()
e = (dG1.float() - dG0.float()).view(N // heads, L, heads, 2, 4, H)
r = dG0.float().view(N // heads, L, heads, 2, 4, H)
print("nan count", int(torch.isnan(dG1.float()).sum()))
rel = lambda a, b: float(a.norm() / (b.norm() + 1e-30))
for d in range(2):
    print("dir", d, "by step:", [f"{rel(e[:, t, :, d], r[:, t, :, d]):.1e}" for t in range(L)])
    print("dir", d, "by gate:", [f"{rel(e[:, :, :, d, k], r[:, :, :, d, k]):.1e}" for k in range(4)])
    print("dir", d, "by 32 units:", [f"{rel(e[..., d, :, u:u + 32], r[..., d, :, u:u + 32]):.1e}" for u in range(0, H, 32)])
en = e.permute(0, 2, 1, 3, 4, 5).reshape(N, -1); rn = r.permute(0, 2, 1, 3, 4, 5).reshape(N, -1)
per = (en.norm(dim=1) / (rn.norm(dim=1) + 1e-30))
print("worst sequences:", [(int(i), f"{float(per[i]):.1e}") for i in per.argsort(descending=True)[:12]])
print("by 16 sequences:", [f"{rel(en[i:i + 16], rn[i:i + 16]):.1e}" for i in range(0, min(N, 256), 16)])
