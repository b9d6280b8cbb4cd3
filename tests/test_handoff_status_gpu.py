"""The product READS the hand-off status words.  The weight-stationary BiLSTM kernels (csrc/lstm_rs.hip,
csrc/lstm_cluster.hip: nn.LSTM of modules/latent.py:87-92) need all their workgroups co-resident; a workgroup whose
peer never arrives gives up after a bounded spin, raises the launch's status word and a sticky per-GPU word, and
finishes on stale tiles.  `dtt_lstm_{rs,cluster}_diag_drop_workgroups(1)` starts the next launch one workgroup short
(what a CU-masked partition or a foreign kernel holding a CU does), and the product must turn that into a DttError:
at once in the blocking check (metrics, checkpoint save, GraphedTrainStep.check), and by the poll after next in the
non-blocking path (loss wrapper, optimizer step, graph replay) -- never silently."""
import pytest
import torch

from oracle.fill import fill_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _short_waits():
    """The product's default wait (~8 s, sized for rank skew) would make every starved launch of this file take that
    long: 2^17 polls (~0.1 s) here."""
    from dtt_amd import ops
    ops.handoff_spin_limit(1 << 17)
    yield
    ops.handoff_spin_limit(0)


def _idp(heads=2, Cc=512):
    from dtt_amd.model import IDPModule
    m = IDPModule(heads, Cc).to(DEV)
    m.load_state_dict(fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 91))
    return m


def _x(B=2, Fq=9, T=5, Cc=512, seed=92):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, Fq, T, Cc, generator=g).to(torch.bfloat16).to(DEV)


@pytest.mark.parametrize("family", ["rs", "cluster"])
def test_a_starved_handoff_launch_raises(family, monkeypatch):
    import dtt_amd
    from dtt_amd import ops
    from dtt_amd._lib import lib, DttError
    monkeypatch.setenv("DTT_LSTM", family)
    drop = getattr(lib(), f"dtt_lstm_{family}_diag_drop_workgroups")
    drop.restype = None
    m, x = _idp(), _x()
    ops.handoff_check()                                   # clean slate
    y0 = m(x)
    ops.handoff_check()                                   # a healthy launch leaves the words at 0
    # blocking path
    drop(1)
    m(x)                                                  # (the output of the starved launch is unspecified)
    with pytest.raises(DttError, match="timed out waiting for a peer workgroup"):
        ops.handoff_check()
    ops.handoff_check()                                   # the failure re-armed the words: clean again
    y2 = m(x)
    ops.handoff_check()
    assert torch.equal(y0, y2)                            # and a healthy launch is exact again
    # non-blocking path: the poll behind the launch queues the read, a later poll raises
    drop(1)
    m(x)
    ops.handoff_poll()
    torch.cuda.synchronize()
    with pytest.raises(DttError, match="INVALID"):
        ops.handoff_poll()
    ops.handoff_check()
    # backward launches are watched too
    xg = x.clone().requires_grad_(True)
    y = m(xg)
    ops.handoff_check()
    drop(1)
    y.float().sum().backward()
    with pytest.raises(DttError):
        ops.handoff_check()


def test_metrics_and_checkpoint_refuse_after_a_starved_launch(tmp_path, monkeypatch):
    """The two places that synchronise anyway read the status: a PIT metric call and save_reference_checkpoint."""
    import dtt_amd
    from dtt_amd import ops
    from dtt_amd._lib import lib, DttError
    monkeypatch.setenv("DTT_LSTM", "rs")
    drop = lib().dtt_lstm_rs_diag_drop_workgroups
    drop.restype = None
    m, x = _idp(), _x()
    ops.handoff_check()
    g = torch.Generator().manual_seed(3)
    audio = torch.randn(2, 800, generator=g).to(DEV)
    srcs = [{"audio": torch.randn(2, 800, generator=g).to(DEV)} for _ in range(2)]
    preds = [{"audio": s["audio"] + 0.1} for s in srcs]
    drop(1)
    m(x)
    with pytest.raises(DttError):
        dtt_amd.SiSnrIPITInference()(audio=audio, preds=preds, sources=srcs)
    drop(1)
    m(x)
    with pytest.raises(DttError):
        dtt_amd.save_reference_checkpoint(str(tmp_path / "ck.pth"), m)
    assert not (tmp_path / "ck.pth").exists()
    dtt_amd.save_reference_checkpoint(str(tmp_path / "ck.pth"), m)      # healthy again
    assert (tmp_path / "ck.pth").exists()


def test_a_starved_launch_inside_a_replayed_graph_raises():
    """A recorded step whose first hand-off launch was captured one workgroup short: every replay times out; the
    replay after next raises from the non-blocking poll, GraphedTrainStep.check() raises at once."""
    import dtt_amd
    from dtt_amd import ops
    from dtt_amd._lib import lib, DttError
    from oracle.fill import synth_sources
    drop = lib().dtt_lstm_rs_diag_drop_workgroups
    drop.restype = None
    # audio-only model whose latent has C' = 256 per head -> H = 512: g = 64 (latent 512 channels), tiny STFT
    model = dtt_amd.DTTNetModel(g=64, n_sources=2, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16, n_heads=2,
                                precision="bf16").to(DEV)
    opt = dtt_amd.ClipAdamW(model.parameters(), lr=1e-4, max_norm=2.0, capturable=True)
    crit = dtt_amd.PITLossWrapper(dtt_amd.DTTNetLoss())
    s1, s2 = synth_sources(2, 800, seed=5)
    inp = {"s1": s1.to(DEV), "s2": s2.to(DEV)}
    state = {"armed": False}

    def batch_fn(i):
        if state["armed"] and torch.cuda.is_current_stream_capturing():
            drop(1)                                       # the next hand-off launch is RECORDED one workgroup short
        s1, s2 = i["s1"], i["s2"]
        mix = s1 + s2
        spec, phase = dtt_amd.stft_mag_phase(mix, 64, 16)
        sources = []
        for s in (s1, s2):
            sp, ph = dtt_amd.stft_mag_phase(s, 64, 16)
            sources.append({"audio": s, "spectrogram": sp, "phase": ph})
        return {"audio": mix, "spectrogram": spec, "phase": phase, "sources": sources, "audio_length": 800}

    ops.handoff_check()
    healthy = dtt_amd.GraphedTrainStep(model, crit, opt, inp, batch_fn, warmup=1)
    for _ in range(3):
        healthy(inp)
    healthy.check()                                       # status words stay 0 across healthy replays
    torch.cuda.current_stream().synchronize()
    for _ in range(2):
        healthy(inp)
    healthy.check()                                       # ... and after a host-side stream sync (DESIGN finding 34)
    state["armed"] = True
    opt2 = dtt_amd.ClipAdamW(model.parameters(), lr=1e-4, max_norm=2.0, capturable=True)
    starved = dtt_amd.GraphedTrainStep(model, crit, opt2, inp, batch_fn, warmup=1)
    state["armed"] = False
    ops.handoff_check()                                   # recording executed nothing
    starved(inp)
    with pytest.raises(DttError):
        starved.check()
    starved(inp)                                          # queues the read behind the replay ...
    torch.cuda.synchronize()
    with pytest.raises(DttError):
        starved(inp)                                      # ... the next call finds it
    try:
        ops.handoff_check()                               # drain what the last replay raised
    except DttError:
        pass
    ops.handoff_check()
