"""Phase timeline of one wave of the register-stationary LSTM kernel (diagnostic build with -DRS_STAMP=<block>):
   DTT_HIP_LIB=scratch/libdtt_dbg.so python tools/lstm_rs_stamps.py N L H"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-separation_amd"))
from dtt_amd._lib import lib, check, ptr, stream_ptr
N, L, H = (int(a) for a in sys.argv[1:4])
which = sys.argv[4] if len(sys.argv) > 4 else "fwd"
dev, dt, heads = "cuda", torch.bfloat16, 2
R = N * L; NT = (N + 31) // 32
xg = (torch.randn(R, 8 * H, device=dev) * 0.7).to(dt); whh = torch.randn(2, 4 * H, H, device=dev) / H ** 0.5
wt = whh.to(dt).contiguous(); whhT = whh.transpose(1, 2).to(dt).contiguous()
dy = torch.randn(R, 2 * H, device=dev).to(dt)
y1 = torch.empty(R, 2 * H, device=dev, dtype=dt)
g1 = torch.empty(2, NT, L, 4 * H * 32, device=dev, dtype=dt); c1 = torch.empty(2, NT, L, H * 32, device=dev)
dG1 = torch.empty(R, 2, 4 * H, device=dev, dtype=dt); dcb = torch.empty(2, NT, H * 32, device=dev); db = torch.zeros(2 * 4 * H, device=dev)
sync = torch.empty(lib().dtt_lstm_rs_sync_words(N, H), device=dev, dtype=torch.int32)
def f(): check(lib().dtt_lstm_rs_fwd(ptr(xg), ptr(wt), ptr(y1), ptr(g1), ptr(c1), ptr(sync), N, L, heads, H, 1, stream_ptr()), "rs fwd")
def b(): check(lib().dtt_lstm_rs_bwd(ptr(dy), ptr(g1), ptr(c1), ptr(whhT), ptr(dG1), ptr(dcb), ptr(db), ptr(sync), N, L, heads, H, 1, stream_ptr()), "rs bwd")
for _ in range(5): f()
fn = f if which == "fwd" else b
for _ in range(5): fn()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (128 * 16))()
check(lib().dtt_lstm_rs_read_stamps(buf, 128 * 16), "stamps")
st = torch.tensor(list(buf), dtype=torch.int64).view(128, 16)
rows = [r for r in range(128) if int(st[r, 0]) > 0]
print(f"{which} N={N} L={L} H={H}: items with stamps {len(rows)}; cycles since the item's first stamp, then item period")
prev = None
for r in rows[-24:]:
    t0 = int(st[r, 0])
    rel = [int(st[r, i]) - t0 if int(st[r, i]) else -1 for i in range(1, 11)]
    print(r, rel, "period", (t0 - prev) if prev else 0)
    prev = t0
