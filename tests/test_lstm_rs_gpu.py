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
