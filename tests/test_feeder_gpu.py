"""Device batch producer (dtt_amd.collate_on_device) against the reference's per-item STFT + collate_fn,
restated in oracle.collate (collate.py:4-83, base_dataset.py:72-138,154-167): ragged lengths, zero padding,
mouth crops as uint8."""
import pytest
import torch

import oracle
from oracle.fill import synth_sources

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _items(lengths, n_fft, hop, with_video, seed=4):
    g = torch.Generator().manual_seed(seed)
    items = []
    for i, L in enumerate(lengths):
        s1, s2 = synth_sources(1, L, seed + i)
        mix = s1 + s2
        it = {"audio": mix, "audio_original": mix.clone(), "audio_path": f"mix_{i}.wav", "sources": []}
        it["spectrogram"], it["phase"] = oracle.stft_mag_phase(mix, n_fft, hop)
        for s in (s1, s2):
            sp, ph = oracle.stft_mag_phase(s, n_fft, hop)
            it["sources"].append({"audio": s, "spectrogram": sp, "phase": ph})
        if with_video:
            it["mouth1"] = torch.randint(0, 256, (5, 8, 8), generator=g, dtype=torch.uint8)
            it["mouth2"] = None if i == 1 else torch.randint(0, 256, (5, 8, 8), generator=g, dtype=torch.uint8)
        items.append(it)
    return items


def _cplx(mag, ph):
    return torch.polar(mag.double().cpu(), ph.double().cpu())


@pytest.mark.parametrize("lengths", [[800, 800, 800], [800, 523, 1024, 523]])
def test_collate_on_device_matches_reference_collate(lengths):
    import dtt_amd
    n_fft, hop = 64, 16
    items = _items(lengths, n_fft, hop, with_video=True)
    ref = oracle.collate(items)
    wave_only = [{k: v for k, v in it.items() if k not in ("spectrogram", "phase")} for it in items]
    for it in wave_only:
        it["sources"] = [{"audio": s["audio"]} for s in it["sources"]]
    got = dtt_amd.collate_on_device(wave_only, DEV, n_fft, hop)
    assert got["audio_length"] == ref["audio_length"] and got["audio_lengths"] == ref["audio_lengths"]
    assert torch.equal(got["spectrogram_length"], ref["spectrogram_length"])
    assert torch.equal(got["audio"].cpu(), ref["audio"])
    assert got["audio_path"] == [f"mix_{i}.wav" for i in range(len(lengths))]
    # phase flips between +pi and -pi where a bin is real up to rounding (frame 0 of the centred STFT):
    # compare the complex spectrum; magnitudes directly
    pairs = [(got, ref)] + list(zip(got["sources"], ref["sources"]))
    for a, b in pairs:
        assert a["spectrogram"].shape == b["spectrogram"].shape
        assert float((a["spectrogram"].cpu() - b["spectrogram"]).abs().max()) <= 2e-5 * float(b["spectrogram"].abs().max())
        za, zb = _cplx(a["spectrogram"], a["phase"]), _cplx(b["spectrogram"], b["phase"])
        assert float((za - zb).abs().max()) <= 2e-5 * float(zb.abs().max())
    for i, L in enumerate(lengths):                           # frames beyond an item's own length are zero
        T = 1 + L // hop
        assert float(got["spectrogram"][i, :, T:].abs().sum()) == 0.0 and float(got["phase"][i, :, T:].abs().sum()) == 0.0
    assert got["video"].dtype == torch.float32 and got["video"].shape == (len(lengths), 2, 5, 8, 8)
    assert torch.allclose(got["video"].cpu(), ref["video"].float() / 255.0, atol=1e-7)
    assert float(got["video"][1, 1].abs().sum()) == 0.0       # missing mouth crop -> zeros (collate.py:41-47)


def test_collated_batch_feeds_the_model():
    import dtt_amd
    items = _items([800, 640], 64, 16, with_video=False)
    for it in items:
        it.pop("spectrogram"), it.pop("phase")
    batch = dtt_amd.collate_on_device(items, DEV, 64, 16)
    model = dtt_amd.DTTNetModel(g=8, n_sources=2, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16, n_heads=2,
                                precision="fp32").to(DEV)
    batch.update(model(**batch))
    res = dtt_amd.PITLossWrapper(dtt_amd.DTTNetLoss())(**batch)
    assert torch.isfinite(res["loss"]).item() and res["batch_permuts"].shape == (2, 2)


def test_spec_augment_on_device_properties():
    """On-device SpecAugment stripes (spec_augs/spec_augmentation.py:7-17 over torchlibrosa's DropStripes; the
    library is absent: parity unpinned, its published algorithm restated).  Checked by what the algorithm guarantees:
    only whole rows / columns are zeroed, at most stripes_num stripes per axis and item each narrower than
    drop_width, everything else untouched, p = 0 is the identity, p = 1 touches every item, deterministic per seed."""
    import dtt_amd
    B, Fq, T = 8, 513, 126
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.rand(B, Fq, T, device=DEV) + 0.5                       # strictly positive: zeros are stripes
    y = dtt_amd.spec_augment_on_device(x, p=1.0, generator=g)
    z = (y == 0)
    assert torch.equal(torch.where(z, x, y), x)                      # untouched elsewhere
    for b in range(B):
        rows = z[b].all(dim=1)                                       # zeroed F rows (torchlibrosa's "time" stripes)
        cols = z[b].all(dim=0)
        assert torch.equal(z[b], rows[:, None] | cols[None, :])      # only whole rows / columns
        for v in (rows, cols):
            d = torch.diff(torch.cat([torch.zeros(1, device=DEV), v.float(), torch.zeros(1, device=DEV)]))
            starts, ends = (d == 1).nonzero().flatten(), (d == -1).nonzero().flatten()
            assert len(starts) <= 2 and all(int(e - s) < 32 for s, e in zip(starts, ends))   # 2 stripes of < 16 may merge
    assert torch.equal(dtt_amd.spec_augment_on_device(x, p=0.0, generator=g), x)
    g1 = torch.Generator(device=DEV).manual_seed(9)
    g2 = torch.Generator(device=DEV).manual_seed(9)
    assert torch.equal(dtt_amd.spec_augment_on_device(x, generator=g1), dtt_amd.spec_augment_on_device(x, generator=g2))
