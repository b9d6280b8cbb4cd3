"""Register-stationary BiLSTM kernels (csrc/lstm_rs.hip: dtt_lstm_rs_fwd / dtt_lstm_rs_bwd) through the C ABI against
the streaming kernels (dtt_lstm_fwd / dtt_lstm_bwd, themselves pinned to the reference's nn.LSTM,
src/model/DTTNet/modules/latent.py:87-92, by the golden IDP fixtures and the CPU oracle) on the same seeded inputs:
output h, saved gates and cell states, gate gradients and the bias gradient, bf16 tolerance.  The shapes walk every
kernel variant: clusters with 1 / 2 tiles (serial form), 3 tiles (item pipeline), >= 4 tiles (deep forward kernel,
every memory operation inside the MFMA loop), ragged last tiles, 1 / 2 / 4 heads, H = 512 and H = 256, and the cfg4
production shapes (N = 2080 x L = 16 time axis, N = 512 x L = 65 frequency axis)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("lstm_rs_check", os.path.join(ROOT, "tools", "lstm_rs_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("N,L,heads,H", [
    (64, 3, 2, 512), (70, 7, 2, 512), (64, 33, 2, 512),            # one tile per cluster, ragged tile
    (1024, 5, 2, 512), (1000, 4, 1, 512),                          # two tiles per cluster
    (1536, 6, 4, 512),                                             # three: item pipeline
    (2080, 16, 2, 512), (512, 65, 2, 512),                         # cfg4: deep kernels / latency-bound chain
    (2300, 5, 2, 512),                                             # deep, ragged, clusters of 4 and 5 tiles
    (130, 6, 2, 256), (2048, 5, 2, 256), (8320, 16, 2, 256),       # H = 256 (2-CU forward, 4-CU backward clusters)
])
def test_lstm_rs_matches_streaming_kernels(N, L, heads, H):
    assert _tool().run(N, L, heads, H)


def test_lstm_rs_refuses_what_it_does_not_cover():
    from dtt_amd._lib import lib
    assert lib().dtt_lstm_rs_supported(2080, 512, 1) == 0
    assert lib().dtt_lstm_rs_supported(2080, 512, 0) == 1          # fp32: the streaming / cluster kernels
    assert lib().dtt_lstm_rs_supported(2080, 1024, 1) == 1
    assert lib().dtt_lstm_rs_sync_words(2080, 384) == 0


@pytest.mark.parametrize("N,L", [(2080, 16), (512, 65)])
def test_lstm_rs_production_shape_against_the_oracle(N, L):
    """cfg4's two recurrent shapes (time axis N = 2080 x L = 16, frequency axis N = 512 x L = 65; H = 512, 2 heads)
    DIRECTLY against the CPU oracle's recurrence (oracle.lstm_dir = the loop inside oracle.lstm_bidir, which the
    reference's nn.LSTM fixtures pin; modules/latent.py:87-92) in fp64 on the same bf16-rounded inputs: outputs and
    gate gradients <= 1e-2 relative, and a per-hidden-unit / per-gate-column bound so that one wrong wave (a
    hidden-unit group of one cluster member) cannot hide inside a norm.  The hand-off status words must read 0."""
    import ctypes as C
    import torch
    import oracle
    from dtt_amd._lib import lib, check, ptr, stream_ptr, dtype_code
    heads, H, dev, dt = 2, 512, "cuda", torch.bfloat16
    R, d, NT = N * L, dtype_code(torch.bfloat16), (N + 31) // 32
    g = torch.Generator().manual_seed(1234 + N)
    xg = (torch.randn(R, 2, 4 * H, generator=g) * 0.7).to(dt)
    whh = (torch.randn(2, 4 * H, H, generator=g) / H ** 0.5).to(dt)
    dy = torch.randn(R, 2 * H, generator=g).to(dt)

    def to_seq(t):         # rows (outer, step, head) -> [N = outer * heads + head, L, ...]
        return t.reshape(N // heads, L, heads, *t.shape[1:]).transpose(1, 2).reshape(N, L, *t.shape[1:])

    def to_rows(t):
        return t.reshape(N // heads, heads, L, *t.shape[2:]).transpose(1, 2).reshape(R, *t.shape[2:])

    # oracle, fp64
    xin = to_seq(xg.double()).requires_grad_(True)                       # [N, L, 2, 4H]
    yo = torch.cat([oracle.lstm_dir(xin[:, :, di], whh[di].double(), bool(di)) for di in (0, 1)], -1)   # [N, L, 2H]
    yo.backward(to_seq(dy.double()))
    y_ref, dG_ref = to_rows(yo.detach()), to_rows(xin.grad)              # [R, 2H], [R, 2, 4H]

    xg_d, whh_d, dy_d = xg.to(dev).reshape(R, 8 * H), whh.to(dev).contiguous(), dy.to(dev)
    y = torch.full((R, 2 * H), float("nan"), device=dev, dtype=dt)
    gates = torch.empty(2, NT, L, 4 * H * 32, device=dev, dtype=dt)
    cst = torch.empty(2, NT, L, H * 32, device=dev)
    nsync = lib().dtt_lstm_rs_sync_words(N, H)
    sf = torch.empty(nsync, device=dev, dtype=torch.int32)
    sb = torch.empty(nsync, device=dev, dtype=torch.int32)
    check(lib().dtt_lstm_rs_fwd(ptr(xg_d), ptr(whh_d), ptr(y), ptr(gates), ptr(cst), ptr(sf), N, L, heads, H, d,
                                stream_ptr()), "dtt_lstm_rs_fwd")
    whhT = whh_d.transpose(1, 2).contiguous()
    dG = torch.full((R, 2, 4 * H), float("nan"), device=dev, dtype=dt)
    dcb = torch.empty(2, NT, H * 32, device=dev)
    db = torch.zeros(2 * 4 * H, device=dev)
    check(lib().dtt_lstm_rs_bwd(ptr(dy_d), ptr(gates), ptr(cst), ptr(whhT), ptr(dG), ptr(dcb), ptr(db), ptr(sb), N, L,
                                heads, H, d, stream_ptr()), "dtt_lstm_rs_bwd")
    torch.cuda.synchronize()
    assert int(sf[0]) == 0 and int(sb[0]) == 0, "hand-off timed out"
    y_, dG_ = y.double().cpu(), dG.double().cpu()
    ey = float((y_ - y_ref).norm() / y_ref.norm())
    eg = float((dG_ - dG_ref).norm() / dG_ref.norm())
    # per column (hidden unit x direction; gate x unit x direction): error norm against that column's own norm
    cy = ((y_ - y_ref).norm(dim=0) / y_ref.norm(dim=0)).max()
    cg = ((dG_ - dG_ref).reshape(R, -1).norm(dim=0) / dG_ref.reshape(R, -1).norm(dim=0)).max()
    my = float((y_ - y_ref).abs().max())                                  # |h| <= 1: bf16 output rounding is 4e-3
    mg = float((dG_ - dG_ref).abs().max() / dG_ref.abs().max())
    edb = float((db.double().cpu() - dG_ref.sum(0).reshape(-1)).norm() / dG_ref.sum(0).norm())
    print(f"N={N} L={L}: y {ey:.2e} (worst column {float(cy):.2e}, max abs {my:.3f})  dG {eg:.2e} (worst column "
          f"{float(cg):.2e}, max {mg:.3f} of the largest)  dbias {edb:.2e}")
    assert ey < 1e-2 and eg < 1e-2, (ey, eg)
    assert float(cy) < 2e-2 and float(cg) < 3e-2, (float(cy), float(cg))
    assert my < 0.03 and mg < 0.03, (my, mg)
    assert edb < 1e-2, edb
